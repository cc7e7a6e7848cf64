// Integration test for blyssprivacy/sdk's lib/spiral-rs: does the reference's own `server::process_query` return, byte
// for byte, the responses this repository's oracle / HIP path produced for the same serialized inputs?
//
// The inputs are written by scripts/ref_check/dump_cases.py of the MI355X port (params.json, pp.bin, query.bin, db.bin,
// response.bin per case).  Copy this file to lib/spiral-rs/tests/ref_check.rs and run
//
//     REF_CHECK_DIR=/path/to/ref_check cargo test --release --features server --test ref_check -- --nocapture
//
// It uses nothing but the crate's public API: util::params_from_json (util.rs:219), PublicParameters::deserialize
// (client.rs:212), Query::deserialize (client.rs:303), server::load_preprocessed_db_from_file (server.rs:373),
// server::expand_query (server.rs:525), server::multiply_reg_by_database (server.rs:155) and server::process_query
// (server.rs:650).  STAGES are compared before the whole query, so that a mismatch names where the two part ways:
//   1. expand_query            -> v_reg.bin (v_reg_reoriented) and v_folding.bin: covers Query::deserialize's ChaCha20 row-0
//                                 regeneration (client.rs:47-49, 303-329), coefficient_expansion, regev_to_gsw
//   2. multiply_reg_by_database on the first database slice -> plane0.bin
//   3. process_query           -> response.bin (fold, pack, encode on top of 1 and 2)
// Run it twice:
//     REF_CHECK_DIR=... cargo test --release --features server --test ref_check -- --nocapture
//     REF_CHECK_DIR=... RUSTFLAGS="-C target-feature=+avx2" cargo test --release --features server --test ref_check -- --nocapture
// The scalar and the AVX2 builds must both agree with the dump.  The AVX2 bodies differ from the scalar ones in
// REPRESENTATIVES, not in residues: `ntt_forward` corrects with strict compares (ntt.rs:163, 193-207) and leaves q where the
// scalar body leaves 0, `multiply` accumulates unreduced (poly.rs:407-426, 460-481).  The stage outputs of expand_query are
// therefore reduced limb by limb before they are compared (`canon_word`, `canon_ntt`); multiply_reg_by_database and the
// response are compared raw (both builds reduce them).  This repository's oracle restates both sets of bodies and checks
// exactly these equalities (tests/test_oracle_avx2_bodies.py), so both runs are expected to pass.
use std::fs::{self, File};
use std::path::{Path, PathBuf};

use spiral_rs::client::{PublicParameters, Query};
use spiral_rs::poly::{PolyMatrix, PolyMatrixNTT};
use spiral_rs::server::{expand_query, load_preprocessed_db_from_file, multiply_reg_by_database, process_query};
use spiral_rs::util::params_from_json;

fn read_words(p: &Path) -> Vec<u64> {
    let b = fs::read(p).unwrap();
    assert_eq!(b.len() % 8, 0, "{:?}: not a whole number of u64 words", p);
    b.chunks_exact(8).map(|c| u64::from_ne_bytes(c.try_into().unwrap())).collect()
}

// both limbs of a packed word lo | hi << 32 (server.rs:262-270, util.rs:343-350) reduced mod their prime
fn canon_word(w: u64, q0: u64, q1: u64) -> u64 {
    ((w & 0xFFFF_FFFF) % q0) | (((w >> 32) % q1) << 32)
}

// an NTT-form polynomial matrix [poly][crt][z]: every residue reduced mod its prime
fn canon_ntt(words: &mut [u64], poly_len: usize, moduli: &[u64]) {
    let per_poly = poly_len * moduli.len();
    for (i, w) in words.iter_mut().enumerate() {
        *w %= moduli[(i % per_poly) / poly_len];
    }
}

fn first_diff(a: &[u64], b: &[u64]) -> Option<usize> {
    if a.len() != b.len() {
        return Some(a.len().min(b.len()));
    }
    a.iter().zip(b.iter()).position(|(x, y)| x != y)
}

fn case_dirs(root: &Path) -> Vec<PathBuf> {
    let mut v: Vec<PathBuf> = fs::read_dir(root)
        .expect("REF_CHECK_DIR is not a directory")
        .filter_map(|e| e.ok())
        .map(|e| e.path())
        .filter(|p| p.join("params.json").is_file())
        .collect();
    v.sort();
    v
}

#[test]
fn reference_process_query_matches_dumped_responses() {
    let root = PathBuf::from(std::env::var("REF_CHECK_DIR").expect("set REF_CHECK_DIR to the directory dump_cases.py wrote"));
    let dirs = case_dirs(&root);
    assert!(!dirs.is_empty(), "no cases under {:?}", root);
    for d in dirs {
        let name = d.file_name().unwrap().to_string_lossy().to_string();
        let params = params_from_json(&fs::read_to_string(d.join("params.json")).unwrap());
        if params.version != 0 {
            // packing version 1 is implemented by lib/server (src/compute/pack.rs:46-99), not by spiral_rs::server::pack
            println!("{}: skipped (params.version = {}: not a spiral_rs::server::process_query case)", name, params.version);
            continue;
        }
        let pp_bytes = fs::read(d.join("pp.bin")).unwrap();
        let q_bytes = fs::read(d.join("query.bin")).unwrap();
        let want = fs::read(d.join("response.bin")).unwrap();
        assert_eq!(pp_bytes.len(), params.setup_bytes(), "{}: pp.bin length", name);
        assert_eq!(q_bytes.len(), params.query_bytes(), "{}: query.bin length", name);
        let pp = PublicParameters::deserialize(&params, &pp_bytes);
        let query = Query::deserialize(&params, &q_bytes);
        let mut f = File::open(d.join("db.bin")).unwrap();
        let db = load_preprocessed_db_from_file(&params, &mut f);
        // ---- stage 1: expand_query (only configurations that expand on the server)
        if params.expand_queries && d.join("v_reg.bin").is_file() {
            let (v_reg, v_folding) = expand_query(&params, &pp, &query);
            let want_reg = read_words(&d.join("v_reg.bin"));
            let (q0, q1) = (params.moduli[0], params.moduli[1]);
            let got_reg: Vec<u64> = v_reg.as_slice().iter().map(|&w| canon_word(w, q0, q1)).collect();
            let diff = first_diff(&got_reg, &want_reg);
            assert!(diff.is_none(), "{}: STAGE expand_query: v_reg_reoriented differs at word {:?}", name, diff);
            let want_fold = read_words(&d.join("v_folding.bin"));
            let mut got_fold: Vec<u64> = Vec::with_capacity(want_fold.len());
            for m in v_folding.iter() {
                got_fold.extend_from_slice(m.as_slice());
            }
            canon_ntt(&mut got_fold, params.poly_len, &params.moduli[0..params.crt_count]);
            let diff = first_diff(&got_fold, &want_fold);
            assert!(diff.is_none(), "{}: STAGE expand_query: v_folding differs at word {:?}", name, diff);
            // ---- stage 2: multiply_reg_by_database on the first (instance 0, trial 0) slice
            let dim0 = 1usize << params.db_dim_1;
            let num_per = 1usize << params.db_dim_2;
            let slice = dim0 * num_per * params.poly_len;
            let mut inter: Vec<PolyMatrixNTT> = (0..num_per).map(|_| PolyMatrixNTT::zero(&params, 2, 1)).collect();
            multiply_reg_by_database(&mut inter, &db.as_slice()[0..slice], v_reg.as_slice(), &params, dim0, num_per);
            let want_plane = read_words(&d.join("plane0.bin"));
            let mut got_plane: Vec<u64> = Vec::with_capacity(want_plane.len());
            for m in inter.iter() {
                got_plane.extend_from_slice(m.as_slice());
            }
            let diff = first_diff(&got_plane, &want_plane);
            assert!(diff.is_none(), "{}: STAGE multiply_reg_by_database: output differs at word {:?}", name, diff);
            println!("{}: expand_query and multiply_reg_by_database identical", name);
        }
        // ---- stage 3: the whole query
        let got = process_query(&params, &pp, &query, db.as_slice());
        assert_eq!(got.len(), want.len(), "{}: response length", name);
        let diff = got.iter().zip(want.iter()).position(|(a, b)| a != b);
        assert!(diff.is_none(), "{}: STAGE fold/pack/encode: response differs from the dumped one at byte {:?}", name, diff);
        println!("{}: {} response bytes identical", name, got.len());
    }
}
