// Integration test for blyssprivacy/sdk's lib/spiral-rs: does the reference's own `server::process_query` return, byte
// for byte, the responses this repository's oracle / HIP path produced for the same serialized inputs?
//
// The inputs are written by scripts/ref_check/dump_cases.py of the MI355X port (params.json, pp.bin, query.bin, db.bin,
// response.bin per case).  Copy this file to lib/spiral-rs/tests/ref_check.rs and run
//
//     REF_CHECK_DIR=/path/to/ref_check cargo test --release --features server --test ref_check -- --nocapture
//
// It uses nothing but the crate's public API: util::params_from_json (util.rs:219), PublicParameters::deserialize
// (client.rs:212), Query::deserialize (client.rs:303), server::load_preprocessed_db_from_file (server.rs:373) and
// server::process_query (server.rs:650).  Run it with and without `RUSTFLAGS="-C target-feature=+avx2"`: the scalar and
// the AVX2 builds must both agree (their intermediates differ, their response bytes must not).
use std::fs::{self, File};
use std::path::{Path, PathBuf};

use spiral_rs::client::{PublicParameters, Query};
use spiral_rs::server::{load_preprocessed_db_from_file, process_query};
use spiral_rs::util::params_from_json;

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
        let got = process_query(&params, &pp, &query, db.as_slice());
        assert_eq!(got.len(), want.len(), "{}: response length", name);
        let first_diff = got.iter().zip(want.iter()).position(|(a, b)| a != b);
        assert!(first_diff.is_none(), "{}: response differs from the dumped one at byte {:?}", name, first_diff);
        println!("{}: {} response bytes identical", name, got.len());
    }
}
