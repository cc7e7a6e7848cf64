#!/usr/bin/env python
"""bench.py -- queries/s of the MI355X Spiral PIR answer path (BASELINE.json metric).

One "step" = one full process_query (Query::deserialize + expand_query + db sweep + fold + pack + encode,
lib/spiral-rs/src/server.rs:650-741) over the resident synthetic database (seed 0x123456789: the database whose
responses tests/test_gpu_fullsize.py compares byte for byte with the oracle).

  --gpus 1 (default)        BASELINE.json configs[1] = C2: 2^20 items x 256 B (nu = (9,11), 64 GiB encoded, 56 GiB
                            resident), ONE query per step.  --batch B: B queries per step sharing database passes.
  --gpus N, mode "shard"    (default for N > 1; north star) the same database ROW-SHARDED over the N GPUs, one process
                            per GPU; every rank expands the query (pruned to its rows), sweeps its dim0/N rows plane by
                            plane, the partial Regev ciphertexts of plane p are reduce-scattered (RCCL ncclSum over xGMI,
                            issued by libspiral_hip.so itself: sp_process_query_sharded) while plane p+1 is swept, every
                            rank folds its num_per/N columns, one all-gather, rank 0 finishes.  Strong scaling.
                            SPIRAL_MULTIGPU=torch|reduce|columns selects the older torch.distributed flows.
  --gpus N, mode "replicas" (BASELINE.json configs[4]; --mode replicas or SPIRAL_BENCH_MODE=replicas) every GPU holds the
                            WHOLE database; a step = 8 N queries (--batch 8 per rank, one database pass per rank);
                            no collective inside the timed region.  Weak scaling.
  --config c3               BASELINE.json configs[2] (2^22 x 256 B; 32 GiB of rows per GPU at N = 8; fits one GPU too).

Prints ONE JSON line on rank 0.  Inputs are resident in HBM before the timed region (the 16 KiB query and the 20 KiB
response are the only host traffic inside a step).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_BASE = {"n": 2, "nu_1": 9, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8, "t_exp_right": 56,
         "instances": 1, "db_item_size": 256}
CONFIGS = {
    # BASELINE.json configs[1]: literal 2^20 x 256 B (SURVEY.md 8(d) "C2"); gadget set of CFG_20_256 (util.rs:7-20)
    "c2": dict(_BASE, nu_2=11),
    # configs[0]: 2^14 x 256 B
    "c1": dict(_BASE, nu_2=5),
    # the reference's own packed preset CFG_20_256 (2^20 x 256 B packed into 2^15 x 8 KiB)
    "p2": dict(_BASE, nu_2=6, db_item_size=8192),
    # configs[3]: 2^20 items x 32 KiB (SpiralWiki-style payload): 16 planes, 256 GiB encoded (224 GiB packed)
    "c4": dict(_BASE, nu_2=11, instances=4, db_item_size=32768),
    # configs[2]: 2^22 items x 256 B (nu = (9,13)); 32 GiB of rows per GPU at --gpus 8, 224 GiB packed on one GPU
    "c3": dict(_BASE, nu_2=13),
    "fast": {"n": 2, "nu_1": 6, "nu_2": 2, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
             "t_exp_right": 8, "instances": 1, "db_item_size": 8192},
}
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
Q = 268369921 * 249561089
SEED = 0x123456789      # util.rs:171-173


def synthetic_wire_bytes(n_bytes, seed):
    """A syntactically valid serialized pp / query: 32-byte seed then LE u64 words < Q.  The answer path's
    arithmetic is data-independent, so throughput on these equals throughput on real ciphertexts; parity on
    real ciphertexts is what tests/ checks."""
    rng = np.random.default_rng(seed)
    body = rng.integers(0, Q, (n_bytes - 32) // 8, dtype=np.uint64)
    return rng.integers(0, 256, 32, dtype=np.uint8).tobytes() + body.tobytes()


def sweep_algorithmic_bytes(cfg, shards):
    """BASELINE.md section 2 / SURVEY.md 8(d): per query, per GPU shard, in the REFERENCE's 8-byte word format."""
    N = 2048
    T = cfg["instances"] * cfg["n"] ** 2
    dim0, num_per = 1 << cfg["nu_1"], 1 << cfg["nu_2"]
    return T * N * num_per * (dim0 // shards) * 8 + T * N * (dim0 // shards) * 2 * 8 + T * num_per * 4 * N * 8


def sweep_moved_bytes(cfg, shards, db_device_bytes):
    """Bytes the sweep has to move in THIS implementation's resident format, per query per shard: the database as
    stored (7-byte PACKED words where applicable) + the query slice (16 B per (z, row)) + the u32 outputs."""
    N = 2048
    T = cfg["instances"] * cfg["n"] ** 2
    dim0, num_per = 1 << cfg["nu_1"], 1 << cfg["nu_2"]
    return db_device_bytes + T * N * (dim0 // shards) * 16 + T * num_per * 4 * N * 4


def pmc_traffic(cfg_name, world, launches, lib_path=None):
    """HBM bytes per sweep launch from the rocprofv3 PMC passes kept under profiles/ (FETCH_SIZE doubled per the
    gfx950 correction + WRITE_SIZE).  bench.py cannot collect counters itself: the figure is REPLAYED from the newest
    tracked record and `traffic_source` says so.  A record is only replayed into a library whose sweep kernel is, byte for
    byte, the kernel that was profiled (`kernel_signature`: sha256 of its machine code, sdk_amd/kernel_signature.py); with a
    different kernel -- or a record without a signature -- the traffic is null and `traffic_source` says why.
    (None, reason) when no matching record exists."""
    if cfg_name != "c2" or world != 1:
        return None, None
    from sdk_amd.kernel_signature import SWEEP_C2, kernel_signature
    try:
        loaded, _ = kernel_signature(lib_path, SWEEP_C2)
    except Exception as e:
        return None, "not replayed: no signature of the loaded library's sweep kernel (%s)" % (repr(e)[:100])
    refused = []
    for name in sorted((n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("pmc_sweep_c2.json")), reverse=True):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
            if rec.get("kernel_signature") != loaded:
                refused.append(name)
                continue
            return (rec["hbm_bytes_per_launch"] * rec.get("launches_per_query", 1) / launches,
                    "replayed from profiles/%s (separate rocprofv3 --pmc passes on the builder's box; not measured in this run; "
                    "the record's kernel signature %s equals the loaded library's)" % (name, loaded))
        except Exception:
            continue
    return None, ("not replayed: the loaded library's sweep kernel (signature %s) is not the kernel any record under profiles/ "
                  "was measured on (%s)" % (loaded, ", ".join(refused) or "no records"))


def torchrun_command(n_gpus, argv, port=None):
    """The command `python bench.py --gpus N ...` turns itself into when no launcher has set WORLD_SIZE: one rank per
    GPU on this node under torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    if port is None:
        port = int(os.environ.get("MASTER_PORT", "0")) or (29500 + os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def gpu_telemetry():
    """Clocks / temperatures / power of GPU 0 from rocm-smi (outside every timed region); {} when the tool is missing."""
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showtemp", "--showpower", "--json"],
                             capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(s in kl for s in ("sclk", "mclk", "fclk", "temperature", "power")):
                keep[k] = v
        return keep
    except Exception as e:   # telemetry only
        return {"error": repr(e)[:120]}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_quota_cpus():
    """CPUs' worth of time the cgroup grants this process per period (cgroup v2 cpu.max, v1 cfs quota); None = unlimited.
    The GPU boxes of round 5 show 256 logical CPUs and allow 16: a team of 128 runs for a fraction of each scheduler period
    and is parked for the rest, which is what made 32 threads "faster" than 64 in round 4's scan."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(cfg_name, cfg):
    """CPU restatement of the reference (oracle/, kind "port") timed on this host, bounded sample, two modes:
    faithful = spiral-rs's own threading (server.rs:682-694: the sweep and the fold of an instance run on ONE thread,
               expansion uses the rayon loops); all_core = lib/server's shape (AVX2 u64-lane sweep,
               lib/server/src/compute/dot_product.rs:59-95, every core offered: z-rows of the sweep and subtrees of the
               fold spread over the threads, lib/server/src/server.rs:53-55).  C1 (configs[0]) is also timed in full.
    EVERY STAGE is timed with its own best team (`team_per_stage`); the all_core value is the sum of the per-stage minima,
    `cores` the largest team any stage used.  r05: (1) the candidates are built around what the process is ALLOWED to use --
    the cgroup's CPU quota (`cpu_quota_cpus`), not the count of logical CPUs it may be scheduled on; (2) every candidate runs
    for at least 0.6 s of wall clock, i.e. several scheduler periods: a 5-ms trial of 128 threads fits into one period's
    quota and measures a burst the process cannot sustain (611 GB/s of sweep on these boxes, profiles/r05_cpu_baseline.md);
    (3) threads are bound one per core and spread over the sockets (OMP_PLACES=cores, OMP_PROC_BIND=spread, set before the
    OpenMP runtime loads) and the restatement's large temporaries come from malloc arenas instead of mmap/munmap per call
    (oracle.orc_tune_allocator)."""
    import oracle
    max_threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    quota = cpu_quota_cpus()
    N = 2048
    if quota is not None and quota < max_threads:
        qn = max(1, int(np.ceil(quota)))
        cands = sorted({t for t in (max(1, qn // 2), qn, 2 * qn, 4 * qn) if t <= max_threads})
    else:
        cands = sorted({max(1, max_threads // d) for d in (8, 4, 2, 1)}) if max_threads >= 8 else [max_threads]
    rng = np.random.default_rng(5)
    MIN_TRIAL_S = float(os.environ.get("SPIRAL_CPU_TRIAL_SECONDS", "0.6"))

    def scan(fn):
        """fn(team) under every candidate team, ascending, each repeated until MIN_TRIAL_S of wall clock have passed (the
        sustained rate under a CPU quota, not a burst); stops once a larger team is 1.5x slower than the best so far.
        -> (best team, {team: seconds per call})"""
        times, best = {}, None
        for t in cands:
            oracle.set_threads(t)
            fn(t)                                   # thread-pool warm-up at this team size (not timed)
            calls, t0 = 0, time.time()
            while calls == 0 or time.time() - t0 < MIN_TRIAL_S:
                fn(t)
                calls += 1
            times[t] = (time.time() - t0) / calls
            if best is None or times[t] < times[best]:
                best = t
            elif times[t] > 1.5 * times[best]:
                break
        return best, times

    def mem_available_gib():
        try:
            for ln in open("/proc/meminfo"):
                if ln.startswith("MemAvailable"):
                    return int(ln.split()[1]) / 2**20
        except OSError:
            pass
        return 0.0

    def one(name, c, full):
        o = oracle.Params(c)
        cl = oracle.Client(o)
        pp = cl.generate_keys(11)
        q = cl.generate_query(12345 % o.num_items, 12)
        dim0, num_per, planes = o.dim0, o.num_per, o.instances * o.n * o.n
        # ---- expansion (+ get_v_folding_neg), in full, per team
        box = {}

        def expand(_t):
            box["v_reg"], box["v_fold"] = o.expand_query(pp, q)
            box["v_neg"] = o.get_v_folding_neg(box["v_fold"])
        team_e, scan_e = scan(expand)
        t_expand = scan_e[team_e]
        v_reg, v_fold, v_neg = box["v_reg"], box["v_fold"], box["v_neg"]
        # ---- sweep.  full: every z-row of every plane (no scaling); sampled: ONE WHOLE PLANE un-sampled in both modes
        # when the host has the memory for it (16 GiB of words at C2), else 2^25 words' worth of z-rows
        plane_gib = N * num_per * dim0 * 8 / 2**30
        whole_plane = (not full) and mem_available_gib() > 2.5 * plane_gib + 8
        nz = N if (full or whole_plane) else max(1, min(N, (1 << 25) // (num_per * dim0)))
        nz1 = nz if (full or whole_plane) else min(nz, 32)
        reps = planes if full else 1
        oracle.set_threads(max_threads)
        dbs = oracle.words_first_touch(nz, num_per * dim0)   # pages placed by the threads that will stream them
        t0 = time.time()
        for _ in range(reps):
            oracle.sweep_rows(dbs[:nz1 * num_per * dim0], v_reg[:nz1 * dim0 * 2], nz1, dim0, num_per)
        t_sweep_1 = (time.time() - t0) * (1 if full else (N / nz1) * planes)
        oracle.sweep_rows_avx2(dbs[:num_per * dim0], v_reg[:dim0 * 2], 1, dim0, num_per)   # thread-pool warm-up
        nzs = min(nz, max(8, (1 << 27) // (num_per * dim0)))   # ~1 GiB of words per trial of the team scan

        def sweep_trial(_t):
            oracle.sweep_rows_avx2(dbs[:nzs * num_per * dim0], v_reg[:nzs * dim0 * 2], nzs, dim0, num_per)
        sweep_trial(0)
        team_s, scan_s = scan(sweep_trial)
        oracle.set_threads(team_s)
        if team_s != max_threads:      # re-place the pages for the team that will actually run
            dbs = oracle.words_first_touch(nz, num_per * dim0)
        passes, t0 = 0, time.time()
        while passes < reps or (not full and time.time() - t0 < MIN_TRIAL_S):   # sampled: several scheduler periods' worth
            oracle.sweep_rows_avx2(dbs, v_reg[:nz * dim0 * 2], nz, dim0, num_per)
            passes += 1
        t_sweep_all = (time.time() - t0) / passes * (reps if full else (N / nz) * planes)
        del dbs
        # ---- fold: a 2^k-leaf subtree (from_ntt of the leaves + 2^k - 1 fold steps), scaled by step count
        k1 = o.db_dim_2 if full else min(o.db_dim_2, 5)
        ka = o.db_dim_2 if full else min(o.db_dim_2, 11)   # all-core: the whole tree of a plane (its serial top included)
        w = 2 * 2 * o.t_gsw * 2 * N
        cts = rng.integers(0, 249561089, (1 << ka) * 2 * 2 * N, dtype=np.uint64)
        oracle.set_threads(1)
        t0 = time.time()
        for _ in range(reps):
            o.from_ntt_fold_parallel(cts[:(1 << k1) * 4 * N], v_fold[:k1 * w], v_neg[:k1 * w], nu=k1, classes=1)
        t_fold_1 = (time.time() - t0) * (1 if full else (num_per / (1 << k1)) * planes)

        def fold_all(t):
            for _ in range(reps):
                o.from_ntt_fold_parallel(cts, v_fold[:ka * w], v_neg[:ka * w], nu=ka, classes=t)
        team_f, scan_f = scan(fold_all)
        t_fold_all = scan_f[team_f] * (1 if full else (num_per / (1 << ka)) * planes)
        # the same two transform-heavy stages with the reference's AVX2 bodies of ntt_forward / ntt_inverse / multiply
        # (ntt.rs:115-210, 260-365; poly.rs:407-481; oracle.avx2_bodies: what a `-C target-cpu=native` build of spiral-rs
        # runs -- same residues, tests/test_oracle_avx2_bodies.py), each at its best team
        def timed(fn, team):
            oracle.set_threads(team)
            fn(team)
            calls, t0 = 0, time.time()
            while calls == 0 or time.time() - t0 < MIN_TRIAL_S:
                fn(team)
                calls += 1
            return (time.time() - t0) / calls
        with oracle.avx2_bodies():
            t_expand_avx2 = timed(expand, team_e)
            t_fold_avx2 = timed(fold_all, team_f) * (1 if full else (num_per / (1 << ka)) * planes)
        oracle.set_threads(max_threads)
        if full:
            how = ("every z-row of all %d planes and the whole fold tree of every plane executed (random residues as "
                   "database words), nothing scaled; pack/encode omitted (<1%%)" % planes)
        else:
            how = ("sweep: %d of %d z-rows of one plane (AVX2, %d threads) / %d of them (scalar u128, 1 thread), x%d planes; "
                   "fold: %d-leaf subtree over %d threads / %d-leaf subtree on 1 thread, scaled to %d leaves x %d planes; "
                   "expand_query + get_v_folding_neg in full (%d threads); pack/encode omitted (<1%%)" %
                   (nz, N, team_s, nz1, planes, 1 << ka, team_f, 1 << k1, num_per, planes, team_e))
        return {"config": name, "sampled": not full,
                "faithful_qps": 1.0 / (t_expand + t_sweep_1 + t_fold_1),
                "all_core_qps": 1.0 / (t_expand + t_sweep_all + t_fold_all),
                "all_core_avx2_ntt_qps": 1.0 / (t_expand_avx2 + t_sweep_all + t_fold_avx2),
                "seconds": {"expand": t_expand, "sweep_1thread": t_sweep_1, "fold_1thread": t_fold_1,
                            "sweep_all_core": t_sweep_all, "fold_all_core": t_fold_all,
                            "expand_avx2_ntt": t_expand_avx2, "fold_all_core_avx2_ntt": t_fold_avx2},
                "team_per_stage": {"expand": team_e, "sweep": team_s, "fold": team_f},
                "team_scan_seconds": {"expand": {str(k): v for k, v in sorted(scan_e.items())},
                                      "sweep_1GiB_trial": {str(k): v for k, v in sorted(scan_s.items())},
                                      "fold": {str(k): v for k, v in sorted(scan_f.items())}},
                "sample": how}

    def cands_global():
        return list(cands)

    main = one(cfg_name, cfg, full=False)
    extra = [one(k, CONFIGS[k], full=True) for k in ("c1", "p2")] if cfg_name not in ("c1", "p2", "fast") else []
    return {
        "value": max(main["all_core_qps"], main["all_core_avx2_ntt_qps"]), "unit": "queries/s", "cores": max(main["team_per_stage"].values()), "kind": "port",
        "host_cpus": os.cpu_count(), "cpu_model": cpu_model(),
        # what the cgroup lets the process use (CPUs' worth of time per scheduler period; None = no quota), and the teams tried
        "cpu_quota_cpus": cpu_quota_cpus(), "teams_tried": cands_global(),
        "team_per_stage": main["team_per_stage"], "team_scan_seconds": main["team_scan_seconds"],
        "modes": {"all_core": main["all_core_qps"], "all_core_avx2_ntt": main["all_core_avx2_ntt_qps"],
                  "faithful": main["faithful_qps"]},
        "seconds_per_query": main["seconds"],
        "sample": "C++ restatement of spiral-rs (oracle/), config %s; value = the faster of all_core (AVX2 u64-lane sweep "
                  "over z-rows as lib/server's dot_product.rs:59-95, fold subtrees in parallel, scalar transform bodies; each "
                  "stage with the team size that sustains the best rate under this process's CPU quota, team_per_stage, "
                  "threads bound one per core) and all_core_avx2_ntt (the same with the reference's AVX2 bodies of ntt_forward "
                  "/ ntt_inverse / multiply, i.e. a target-cpu=native build); faithful mode = spiral-rs threading (sweep + fold "
                  "on one thread per instance, server.rs:682-694). %s" % (cfg_name, main["sample"]),
        "unsampled": extra,
    }


def batched_pass_record(sp, cfg, db, runs, iters):
    """One database pass for a whole group of begun queries, timed alone (sp_bench_sweep_batch: HIP events on the launch stream),
    with the kernel that ran named from sp_paths_taken -- not from what the shape would normally pick -- and its bytes counted
    in the format it READ: the digit-planar copy (8 bytes per word) for k_sweep_planar, the PACKED words (7) otherwise."""
    sp.paths_taken()
    pass_ms = sp.bench_sweep_batch(runs, db, iters)
    taken = sp.paths_taken()
    B = len(runs)
    N_, T_ = 2048, cfg["instances"] * cfg["n"] ** 2
    planar = "sweep_batch_planar" in taken
    two_tiles = "sweep_batch_mfma_two_tiles" in taken
    db_bytes = db.batch_copy_bytes() if planar else db.device_bytes()
    out_bytes = B * T_ * (1 << cfg["nu_2"]) * 4 * N_ * 4
    pass_bytes = db_bytes + B * N_ * (1 << cfg["nu_1"]) * 16 + out_bytes
    kernel = ("k_sweep_planar<4, 2, 0, 1, 8> (two query tiles, digit-planar database)" if planar else
              "k_sweep_mfma_batch<8, 1, 0, 2> (two query tiles, PACKED database)" if two_tiles else
              "k_sweep_mfma_batch<2, 2>" if "sweep_batch_mfma" in taken else "k_sweep_packed_batch<%d>" % B)
    return {"kernel": kernel, "planar": planar, "two_query_tiles": two_tiles, "queries_per_pass": B,
            "database_format": "digit-planar copy, 8 bytes per word (sweep_planar.hpp)" if planar else "PACKED, 7 bytes per word",
            "ms_per_pass": pass_ms, "ms_of_pass_per_query": pass_ms / B, "bytes_per_pass": pass_bytes,
            "achieved": pass_bytes / (pass_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": pass_bytes / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "note": "database in the format the pass read + %d query slices + %d x u32 outputs (%.0f MB of HBM writes inside the "
                    "read stream)" % (B, B, out_bytes / 1e6)}


def batched_step(sp, torch, p, pp, db, cfg, B, steps, iters, single=None):
    """`steps` timed lists of B queries through sp_process_query_batch after a self-check against the single-query path and two
    warm-up lists (the first of which builds whatever the batched call builds once), + the pass kernel's own record."""
    qs = [synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(B)]
    outs = sp.process_query_batch(p, pp, qs, db)
    if single is None:
        single = [sp.process_query(p, pp, q, db) for q in qs]
    check = "ok" if outs == single else "MISMATCH"
    if check != "ok":
        print("bench: responses of the %d-query step DIFFER from the single-query path" % B, file=sys.stderr, flush=True)
    for _ in range(2):
        sp.process_query_batch(p, pp, qs, db)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sp.process_query_batch(p, pp, qs, db)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    runs = [sp.QueryRun(p, pp, q, db=db) for q in qs]
    try:
        rec = batched_pass_record(sp, cfg, db, runs, iters)
    finally:
        for r in runs:
            r.free()
    return {"value": (B * steps / dt) if check == "ok" else None, "unit": "queries/s", "steps": steps, "queries_per_step": B,
            "ms_per_step": dt * 1e3 / steps, "batch_selfcheck": check, "batched_pass": rec}, single


def secondary_same_db(sp, torch, args, p, pp, db, queries, cfg, step):
    """Measurements the driver's default run also records, on the database the headline just used (BASELINE configs[1]
    resident): `sustained` (a long run of consecutive single queries: q/s of the first and of the last 100, GPU
    telemetry before and after) and `batch8` (BASELINE configs[4] at one GPU: 8 queries per database pass)."""
    out = {}
    if args.sustained > 0:
        n = args.sustained
        tel0 = gpu_telemetry()
        stamps = np.zeros(n + 1)
        torch.cuda.synchronize()
        stamps[0] = time.perf_counter()
        for i in range(n):
            step(i)                       # synchronous: returns the response bytes
            stamps[i + 1] = time.perf_counter()
        tel1 = gpu_telemetry()
        k = max(1, min(100, n // 3))
        dt = np.diff(stamps)
        rel = stamps[1:] - stamps[0]
        by_second = [int(((rel > t) & (rel <= t + 1)).sum()) for t in range(int(rel[-1]))]   # completed queries per whole second
        out["sustained"] = {
            "queries": n, "value": n / (stamps[-1] - stamps[0]), "unit": "queries/s",
            "first_%d_qps" % k: k / (stamps[k] - stamps[0]), "last_%d_qps" % k: k / (stamps[-1] - stamps[-1 - k]),
            "queries_completed_per_second": by_second,
            "ms_per_query_p50": float(np.median(dt) * 1e3), "ms_per_query_max": float(dt.max() * 1e3),
            "seconds": float(stamps[-1] - stamps[0]),
            "telemetry_before": tel0, "telemetry_after": tel1,
            "note": "the headline's step repeated back to back, one query at a time, no pause; rocm-smi sampled outside "
                    "the loop"}
    if cfg["nu_2"] >= 7:
        rec8, single8 = batched_step(sp, torch, p, pp, db, cfg, 8, 5, args.sweep_iters)
        out["batch8"] = dict({"workload": "BASELINE configs[4] at one GPU: 8 queries per step sharing ONE database pass "
                                          "(sp_process_query_batch)"}, **rec8)
        # sixteen queries per step: ONE pass for all of them (two query tiles on the matrix cores; over the digit-planar copy of
        # the database where the device has room for one -- built here, at load time, as a host would: sp_db_prepare_batch)
        prepared = db.prepare_batch()
        single16 = single8 + [sp.process_query(p, pp, synthetic_wire_bytes(p.query_bytes(), 100 + i), db) for i in range(8, 16)]
        rec16, _ = batched_step(sp, torch, p, pp, db, cfg, 16, 3, args.sweep_iters, single=single16)
        out["batch16"] = dict({"workload": "16 queries per step sharing ONE database pass (sp_process_query_batch; two query tiles "
                                           "per pass: %s; digit-planar copy built by sp_db_prepare_batch: %s)"
                                           % (rec16["batched_pass"]["two_query_tiles"], prepared)}, **rec16)
        # the same sixteen through the FALLBACK of that pass -- the two-tile kernel over the PACKED words, what databases with no
        # room for a planar copy run (C3, C4) -- so that it has a number too; switching the copy off releases its memory
        sp.lib().sp_debug_set(b"batch_planar", C.c_long(0))
        try:
            recf, _ = batched_step(sp, torch, p, pp, db, cfg, 16, 2, args.sweep_iters, single=single16)
            out["batch16_packed_fallback"] = dict({"workload": "as batch16 with SPIRAL_BATCH_PLANAR=0: k_sweep_mfma_batch, two query "
                                                               "tiles, over the PACKED words"}, **recf)
        finally:
            sp.lib().sp_debug_set(b"batch_planar", C.c_long(1))
    return out


def secondary_c4(sp, torch, args):
    """BASELINE configs[3] (2^20 items x 32 KiB, 16 planes, 224 GiB resident) on a fresh synthetic fill: the caller has
    released the headline's database.  5 timed single queries after 2 warm-up queries."""
    cfg = CONFIGS["c4"]
    free = torch.cuda.mem_get_info()[0]
    if free < 236 * 2**30:
        return {"skipped": "needs 236 GiB of free HBM, %.1f available" % (free / 2**30)}
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, synthetic_wire_bytes(p.setup_bytes(), 1))
    t0 = time.perf_counter()
    db = sp.Database(p).fill_synthetic(SEED)
    torch.cuda.synchronize()
    fill_s = time.perf_counter() - t0
    queries = [synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(4)]
    steps, warm = 5, 2
    planes = cfg["instances"] * cfg["n"] ** 2

    def one(i):
        run = sp.QueryRun(p, pp, queries[i % len(queries)], db=db)
        run.sweep(db)
        o = run.finish()
        t = run.timings()
        run.free()
        return o, t
    for i in range(warm):
        one(i)
    torch.cuda.synchronize()
    stage = np.zeros(4)
    t0 = time.perf_counter()
    for i in range(steps):
        _, t = one(i)
        stage += np.array(t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    launches = sp.lib().sp_sweep_launches(C.c_void_p(p.h), C.c_void_p(db.h))
    moved = sweep_moved_bytes(cfg, 1, db.device_bytes()) / launches
    in_situ_ms = stage[1] / steps / launches
    res = {"workload": "BASELINE configs[3]: spiral-rs process_query on 2^20 items x 32 KiB (%d planes), %.1f GiB resident, "
                       "one query per step" % (planes, db.device_bytes() / 2**30),
           "value": steps / dt, "unit": "queries/s", "steps": steps, "warmup": warm, "ms_per_step": dt * 1e3 / steps,
           "stage_ms": {"expand": stage[0] / steps, "sweep": stage[1] / steps, "fold": stage[2] / steps,
                        "pack_encode": stage[3] / steps},
           "roofline": {"bound": "hbm", "launches_per_query": launches, "bytes_per_launch": moved, "ms_per_launch": in_situ_ms,
                        "achieved": moved / (in_situ_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": moved / (in_situ_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
           "fill_seconds": fill_s}
    # the batched pass on a database too large for a digit-planar copy: 8 queries per step over the PACKED words (VERDICT r05
    # item 4; the two-tile PACKED kernel's number is `secondary.batch16_packed_fallback`, on C2)
    # (groups of 16 do not form here: 32 workspaces of ~3 GiB do not fit beside 224 GiB, the library keeps to groups of 8)
    try:
        res["batch8"], _ = batched_step(sp, torch, p, pp, db, cfg, 8, 2, 2)
    except sp.SpiralError as e:
        res["batch8"] = {"error": str(e)[:300]}
    del db, pp, p
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=os.environ.get("SPIRAL_BENCH_CONFIG", "c2"), choices=sorted(CONFIGS))
    ap.add_argument("--mode", default=os.environ.get("SPIRAL_BENCH_MODE", "auto"), choices=["auto", "shard", "replicas"],
                    help="N > 1: shard (row shards + RCCL exchange, default) or replicas (whole database per GPU, "
                         "queries split over the GPUs, no collective; BASELINE configs[4])")
    ap.add_argument("--sweep-iters", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-in-flight", action="store_true",
                    help="also time the same queries with two in flight (extra object `two_in_flight` in the line; off by "
                         "default so that the kernel statistics of the default command hold the timed steps only)")
    ap.add_argument("--batch", type=int, default=0,
                    help="queries per step per GPU (single-GPU and replicas modes): groups of <= 8 queries share one "
                         "database pass (sp_process_query_batch).  Default 1 (single) / 8 (replicas).")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline timed region + roofline (no `secondary` objects: sustained / batch8 / batch16 / c4); "
                         "what the profiling scripts use so that kernel statistics hold the headline's launches only")
    ap.add_argument("--sustained", type=int, default=1000,
                    help="queries of the `secondary.sustained` run (0 = skip); the default runs for ~12 s at C2: round 3 saw a "
                         "loop lose 8 %% after ten seconds")
    ap.add_argument("--via-torchrun", action="store_true",
                    help="re-exec under torch.distributed.run even for --gpus 1 (what --gpus N > 1 does on its own when "
                         "no launcher set WORLD_SIZE)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.via_torchrun):
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU) instead of exiting
        cmd = torchrun_command(args.gpus, [a for a in sys.argv[1:] if a != "--via-torchrun"])
        print("bench: no WORLD_SIZE in the environment, re-executing as: %s" % " ".join(cmd), file=sys.stderr, flush=True)
        os.execv(cmd[0], cmd)
    try:   # cpu_baseline: every CPU this process may run on
        n_cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n_cpus = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(n_cpus))
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        # cpu_baseline's OpenMP teams: one place per core, threads spread over the sockets and kept there (read when an OpenMP
        # runtime loads; binding also pins THIS thread to the first place, so never with several ranks on one host -- every
        # rank's main thread and the RCCL helpers it spawns would share one core)
        os.environ.setdefault("OMP_PLACES", "cores")
        os.environ.setdefault("OMP_PROC_BIND", "spread")
    else:
        os.environ.setdefault("OMP_PROC_BIND", "false")
    import torch
    import torch.distributed as dist
    import sdk_amd as sp
    from sdk_amd.sharding import (Comm, gather_local, local_cts_tensor, partial_tensor, reduce_partials,
                                  scatter_fold_query)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    if hasattr(sp.lib(), "sp_emulated_device_marker"):   # SPIRAL_HIP_LIB pointing at tests/emu/_build: a test tool, not a device
        raise SystemExit("bench.py measures the gfx950 library; the loaded one is the tests' host emulation")
    if sp.lib().sp_set_device(local_rank) != 0:
        raise SystemExit("sp_set_device failed")
    # SPIRAL_FORCE_DIST=1: run the N > 1 code path (process group, collectives) at world size 1 (1-GPU boxes)
    use_dist = world > 1 or os.environ.get("SPIRAL_FORCE_DIST") == "1"
    replicas = args.mode == "replicas"
    real_stdout = None
    if use_dist:
        # RCCL prints a version banner through C stdio on stdout; keep stdout for the one JSON line
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    cfg = CONFIGS[args.config]
    p = sp.Params(cfg)
    planes = cfg["instances"] * cfg["n"] ** 2
    pp = sp.PublicParameters.deserialize(p, synthetic_wire_bytes(p.setup_bytes(), 1))
    batch = args.batch if args.batch > 0 else (8 if replicas else 1)
    n_q = max(4, batch)
    queries = [synthetic_wire_bytes(p.query_bytes(), 100 + 1000 * (rank if replicas else 0) + i) for i in range(n_q)]
    if replicas or not use_dist:
        mode = "replicas" if replicas else "single"
    else:
        mode = os.environ.get("SPIRAL_MULTIGPU", "lib")   # lib | torch | reduce | columns
        if mode in ("lib", "torch", "columns") and (1 << cfg["nu_2"]) < world:
            mode = "reduce"
    sharded = mode in ("lib", "torch", "reduce", "columns")
    db = sp.Database(p, rank if sharded else 0, world if sharded else 1,
                     by_columns=(mode == "columns")).fill_synthetic(SEED)
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    comm = None
    selfcheck = "not_applicable"
    overlap = os.environ.get("SPIRAL_OVERLAP", "1") != "0"
    if mode == "lib":
        # the library's own communicator: rank 0 makes the RCCL id, torch.distributed (already up for the barrier /
        # timing contract) carries the 128 bytes to the other ranks
        try:
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(Comm.unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, src=0)
            comm = Comm.rccl(rank, world, bytes(idt.cpu().numpy().tobytes()))
        except Exception as e:  # loud, and every rank takes the same decision
            print("bench[rank %d]: sp_comm_create failed: %r" % (rank, e), file=sys.stderr, flush=True)
            comm = None
        ok = torch.tensor([1 if comm is not None else 0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            comm, mode, selfcheck = None, "torch", "lib_comm_unavailable"

    def step(i):
        """-> (response bytes on rank 0 or None, stage timings or None)"""
        if mode in ("single", "replicas") and batch > 1:
            outs = sp.process_query_batch(p, pp, [queries[(i + k) % len(queries)] for k in range(batch)], db)
            return outs[0], None
        if mode == "lib":
            out = comm.process_query(p, pp, queries[i % len(queries)], db)
            return (out if rank == 0 else None), None
        run = sp.QueryRun(p, pp, queries[i % len(queries)], db=db)  # row shards: expansion pruned to the shard's rows
        if mode == "columns":
            # column shards: complete outputs per shard, no partial sums; only the folded cts are gathered
            run.sweep(db)
            run.fold_local(run.partial_ptr(), world)
            run.sync()
            gathered = gather_local(local_cts_tensor(run), rank, world, dst=0)
            torch.cuda.synchronize()
            out = run.finish_gathered(gathered.data_ptr(), world) if rank == 0 else None
        elif mode == "torch":
            out = scatter_fold_query(run, db, rank, world, overlap=overlap)
        elif mode == "reduce":
            run.sweep(db)
            run.sync()
            reduce_partials(partial_tensor(run), dst=0)  # RCCL ncclSum over xGMI onto rank 0
            torch.cuda.synchronize()
            out = run.finish() if rank == 0 else None
        else:
            run.sweep(db)
            out = run.finish()
        t = run.timings() if out is not None else None
        run.free()
        return out, t

    if mode in ("lib", "torch") and (mode == "lib" or overlap):
        # self-check before timing: the stream-ordered flow must reproduce, byte for byte, the host-synchronised
        # reference flow of the same data path (one sweep launch, one reduce-scatter, torch.distributed collectives).
        # A disagreement is reported in the JSON line ("overlap_selfcheck": "fell_back") and the synchronised flow is
        # timed instead -- never silently.
        fast_mode, fast_overlap = mode, overlap
        mode, overlap = "torch", False
        ref, _ = step(0)
        mode, overlap = fast_mode, fast_overlap
        try:
            got, _ = step(0)
            good = rank != 0 or got == ref
        except Exception as e:
            print("bench[rank %d]: %s flow raised %r" % (rank, fast_mode, e), file=sys.stderr, flush=True)
            good = False
        ok = torch.tensor([1 if good else 0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            selfcheck = "ok" if selfcheck == "not_applicable" else selfcheck + "; torch overlapped flow ok"
        else:
            mode, overlap, selfcheck = "torch", False, "fell_back"
            if rank == 0:
                print("bench: the stream-ordered multi-GPU flow DISAGREED with the synchronised flow; timing the latter",
                      file=sys.stderr, flush=True)
    batch_selfcheck = None
    if mode in ("single", "replicas") and batch > 1:
        # the batched entry point must answer every query exactly as the one-at-a-time path does (that path is compared
        # byte for byte with the oracle at this size by tests/test_gpu_fullsize.py); checked before timing, loudly
        outs = sp.process_query_batch(p, pp, [queries[k % len(queries)] for k in range(batch)], db)
        single = [sp.process_query(p, pp, queries[k % len(queries)], db) for k in range(min(batch, len(queries)))]
        batch_selfcheck = "ok" if all(outs[k] == single[k % len(single)] for k in range(batch)) else "MISMATCH"
        if batch_selfcheck != "ok":
            print("bench[rank %d]: batched responses DIFFER from the single-query path" % rank, file=sys.stderr, flush=True)
            raise SystemExit(3)
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    stage = np.zeros(4)
    rank_ms = np.zeros(2)
    for i in range(args.steps):
        out, t = step(i)
        if t is not None:
            stage += np.array(t)
        if mode == "lib":
            rank_ms += np.array(comm.timings()[:2])
    barrier()
    my_elapsed = time.perf_counter() - t0
    elapsed = my_elapsed
    per_rank = None
    if use_dist:
        tt = torch.tensor([my_elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed * 1e3 / args.steps

    # dominant kernel: the db sweep.  HIP events on the launch stream around `sweep_iters` launches.
    # two queries in flight (single-GPU, one query per step): begin + sweep of query k+1 are queued before finish(k) is
    # waited for, so that its expansion runs under query k's sweeps and k's last fold + pack under k+1's first sweep.
    # Reported beside the headline, never as it (the headline stays one query at a time, comparable across rounds).
    in_flight = None
    if mode == "single" and batch == 1 and world == 1 and args.two_in_flight:
        def two_in_flight(n):
            prev = None
            for i in range(n):
                r = sp.QueryRun(p, pp, queries[i % len(queries)], db=db)
                r.sweep(db)
                if prev is not None:
                    prev.finish()
                    prev.free()
                prev = r
            o = prev.finish()
            prev.free()
            return o
        two_in_flight(max(2, args.warmup))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        last = two_in_flight(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        in_flight = {"queries_in_flight": 2, "value": args.steps / dt, "unit": "queries/s", "ms_per_query": dt * 1e3 / args.steps,
                     "responses": "identical" if last == sp.process_query(p, pp, queries[(args.steps - 1) % len(queries)], db) else "DIFFER",
                     "note": "same %d queries through sp_query_begin / sp_query_sweep / sp_query_finish with the next query "
                             "queued before the previous one is waited for; NOT the headline value" % args.steps}
    run = sp.QueryRun(p, pp, queries[0], db=db)
    per_plane_flow = mode in ("lib",) or (mode == "torch" and overlap)
    sp.paths_taken()
    if per_plane_flow:                                        # one launch per plane, exchange overlapped
        sweep_ms, launches = run.bench_sweep(db, args.sweep_iters, per_plane=1), planes
    else:                                                     # 1, or one per plane when the fold is overlapped
        sweep_ms, launches = run.bench_sweep(db, args.sweep_iters), run.sweep_launches(db)
    run.free()
    sweep_paths = sp.paths_taken()
    batch_pass = None
    if mode in ("single", "replicas") and batch > 1 and cfg["nu_2"] >= 7 and (1 << cfg["nu_1"]) % 2 == 0:   # PACKED databases only
        runs = [sp.QueryRun(p, pp, queries[k % len(queries)], db=db) for k in range(min(batch, 16 if batch > 8 else 8))]
        try:
            batch_pass = batched_pass_record(sp, cfg, db, runs, args.sweep_iters)
        except sp.SpiralError:          # a shape without the two-tile pass: groups of 8
            for r in runs[8:]:
                r.free()
            runs = runs[:8]
            batch_pass = batched_pass_record(sp, cfg, db, runs, args.sweep_iters)
        for r in runs:
            r.free()
    shards = world if sharded else 1
    alg_bytes = sweep_algorithmic_bytes(cfg, shards) / launches
    moved_bytes = sweep_moved_bytes(cfg, shards, db.device_bytes()) / launches
    # In the timed region the sweep launches are the only work on the query's main stream between the "expanded" and
    # "swept" HIP events, so that span / launches is the kernel's average duration in the real run -- including what
    # it loses to the folds that share the CUs from the second stream.  (Batched steps have no per-query span.)
    if mode == "single" and batch == 1 and stage[1] > 0:
        in_situ_ms, in_situ_src = stage[1] / args.steps / launches, "HIP events around the sweep launches inside the timed steps"
    elif mode == "lib" and rank_ms[0] > 0:
        in_situ_ms, in_situ_src = rank_ms[0] / args.steps / launches, "HIP events around the per-plane sweep launches inside the timed steps (rank 0)"
    else:
        in_situ_ms, in_situ_src = sweep_ms, "stand-alone launches (no per-query span in this mode)"
    achieved = moved_bytes / (in_situ_ms * 1e-3) / 1e9
    standalone = moved_bytes / (sweep_ms * 1e-3) / 1e9
    if use_dist:
        pr = torch.tensor([my_elapsed * 1e3 / args.steps, sweep_ms * launches, in_situ_ms * launches], device="cuda",
                          dtype=torch.float64)
        allr = [torch.zeros_like(pr) for _ in range(world)]
        dist.all_gather(allr, pr)
        per_rank = {"step_ms": [float(x[0]) for x in allr], "sweep_standalone_ms_per_query": [float(x[1]) for x in allr],
                    "sweep_in_situ_ms_per_query": [float(x[2]) for x in allr]}
        # what a first run on a multi-GPU node needs in its log to be diagnosable: per rank, the
        # library's own account of its communicator (transport, RCCL version, bytes per collective, the last query's exposed
        # exchange) in the sharded mode, the batched pass's own time and HBM fraction in the replicas mode
        mine = {"rank": rank, "device": torch.cuda.current_device(),
                "env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_", "GPU_MAX_HW_QUEUES", "HIP_VISIBLE"))}}
        if mode == "lib":
            try:
                mine["comm"] = comm.describe()
            except Exception as e:          # diagnostics only
                mine["comm"] = {"error": repr(e)[:200]}
        if batch_pass is not None:
            mine["batched_pass"] = {k: batch_pass[k] for k in ("kernel", "queries_per_pass", "ms_per_pass", "achieved", "frac")}
        gathered = [None] * world
        try:
            dist.all_gather_object(gathered, mine)
            per_rank["ranks"] = gathered
        except Exception as e:
            per_rank["ranks"] = [mine, {"error": "all_gather_object: " + repr(e)[:200]}]

    if rank == 0:
        q_per_step = batch * (world if replicas else 1) if mode in ("single", "replicas") else 1
        traffic, traffic_source = pmc_traffic(args.config, world if sharded else 1, launches, sp.library_path())
        # the roofline block describes the kernel sp_bench_sweep / the stage events time: the single-query sweep; batched
        # steps add roofline.batched_pass (the pass kernel of the group, timed by sp_bench_sweep_batch)
        npairs_local = (1 << cfg["nu_1"]) // (2 * (world if sharded and mode != "columns" else 1))
        ring_u = next((u for u in (8, 4, 2) if npairs_local % (2 * u) == 0), 0)
        kernel = ("k_sweep_packed_ring<%d>" % ring_u if "sweep_ring" in sweep_paths else
                  "k_sweep_packed_persist" if "sweep_packed_persist" in sweep_paths else
                  "k_sweep_wide" if cfg["nu_2"] >= 7 else "k_sweep_narrow2")
        workload_how = {
            "single": "unsharded, one query per step" if batch == 1 else "unsharded, %d queries per step (<= 8 per database pass)" % batch,
            "replicas": "whole database on each of the %d GPUs, %d queries per GPU per step (one database pass), no collective in the timed region (BASELINE configs[4])" % (world, batch),
            "lib": "row-sharded dim0/%d per GPU; RCCL reduce-scatter of the partial Regev cts per plane (overlapping the next plane's sweep), distributed fold, all-gather -- issued by libspiral_hip.so (sp_process_query_sharded)" % world,
            "torch": "row-sharded dim0/%d per GPU + RCCL reduce-scatter of partial Regev cts%s, distributed fold, all-gather (torch.distributed)" % (world, " (per plane, overlapping the next plane's sweep)" if overlap else ", host-synchronised"),
            "reduce": "row-sharded dim0/%d per GPU + RCCL reduce onto rank 0" % world,
            "columns": "column-sharded num_per/%d per GPU, distributed fold, all-gather (no partial sums)" % world}[mode]
        line = {
            "metric": "PIR queries/sec (full answer path) on 2^%d items x %d B" % (cfg["nu_1"] + cfg["nu_2"], cfg["db_item_size"]),
            "value": args.steps * q_per_step / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak" if replicas else "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "mode": mode,
            "rccl_ranks": dist.get_world_size() if use_dist else 1,
            "overlap_selfcheck": selfcheck,
            "batch_selfcheck": batch_selfcheck,
            "per_rank": per_rank,
            "config": {"arithmetic": "exact integers: u32 x u32 -> u64 multiply-accumulate mod two 28-bit primes, u32 NTT "
                                     "butterflies (Shoup), no floating point",
                       "workload": "spiral-rs process_query, %s = %s, encoded DB %.1f GiB (%.1f GiB resident per GPU), %s" %
                                   (args.config, json.dumps(cfg, sort_keys=True), p.db_words * 8 / 2**30,
                                    db.device_bytes() / 2**30, workload_how),
                       "queries_per_step": q_per_step,
                       "stage_ms": {"expand": stage[0] / args.steps, "sweep": stage[1] / args.steps,
                                    "fold": stage[2] / args.steps, "pack_encode": stage[3] / args.steps},
                       "rank0_ms": {"sweep_with_overlapped_exchange": rank_ms[0] / args.steps,
                                    "exchange_tail_fold_gather": rank_ms[1] / args.steps} if mode == "lib" else None},
            "roofline": {"bound": "hbm", "kernel": kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "bytes_per_launch": moved_bytes, "ms_per_launch": in_situ_ms, "ms_per_launch_source": in_situ_src,
                         "launches_per_query": launches,
                         "standalone": {"ms_per_launch": sweep_ms, "achieved": standalone, "frac": standalone / HBM_PEAK_GBPS,
                                        "note": "the same launches issued back to back with nothing else on the GPU "
                                                "(sp_bench_sweep, HIP events on the launch stream)"},
                         "reference_format_equivalent": {
                             "bytes_per_launch": alg_bytes, "GBps": alg_bytes / (in_situ_ms * 1e-3) / 1e9,
                             "note": "SURVEY.md 8(d) counts the reference's 8-byte words; the resident database is bit-"
                                     "packed to 7 bytes per word, so this rate is 8/7 of the bytes really moved and is "
                                     "NOT a fraction of the hardware peak"},
                         "note": "achieved = bytes this implementation moves per launch (PACKED 7-byte database words + "
                                 "query slice + u32 outputs; equals the PMC-measured traffic) / average duration of a "
                                 "sweep launch; peak = HBM3E spec; a plain device copy reaches ~6300 GB/s on this part"},
        }
        if batch_pass is not None:
            line["roofline"]["batched_pass"] = batch_pass
        if in_flight is not None:
            line["two_in_flight"] = in_flight
        if mode == "single" and batch == 1 and world == 1 and not args.headline_only:
            # after the headline (whose timed region and roofline are complete above): the BASELINE configs only the
            # builder had timed so far, now in the line the driver records.  A failure here never takes the headline down.
            secondary = {}
            try:
                secondary.update(secondary_same_db(sp, torch, args, p, pp, db, queries, cfg, step))
            except Exception as e:
                secondary["error_same_db"] = repr(e)[:300]
            if args.config == "c2":
                try:
                    import gc
                    run = None                 # (a freed QueryRun still references its Params / PublicParameters)
                    del db, pp, p
                    gc.collect()
                    torch.cuda.synchronize()
                    secondary["c4"] = secondary_c4(sp, torch, args)
                except Exception as e:
                    secondary["c4"] = {"error": repr(e)[:300]}
            line["secondary"] = secondary
        if mode == "single" and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config, cfg)
        else:
            line["cpu_baseline"] = None
        if real_stdout is not None:
            C.CDLL(None).fflush(None)
            sys.stdout.flush()
            os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        if real_stdout is not None:
            os.dup2(2, 1)
    if comm is not None:
        comm.free()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
