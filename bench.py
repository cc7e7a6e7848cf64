#!/usr/bin/env python
"""bench.py -- queries/s of the MI355X Spiral PIR answer path (BASELINE.json metric).

One "step" = one full process_query (Query::deserialize + expand_query + db sweep + fold + pack +
encode, server.rs:650-741) of ONE query over the resident synthetic database.
  N = 1 : BASELINE.json configs[1] -- 2^20 items x 256 B (nu = (9,11), 64 GiB encoded DB in HBM).
  N > 1 : the same database row-sharded over the N GPUs (dim0/N first-dimension rows each); every rank
          sweeps its shard, the partial Regev ciphertexts (u32 residues) are summed onto rank 0 with one
          RCCL reduce over xGMI, rank 0 folds/packs/encodes.  Strong scaling: total work is fixed.
Prints ONE JSON line on rank 0.  Inputs are resident in HBM before the timed region (the 16 KiB query and
the 20 KiB response are the only host traffic inside a step).
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

CONFIGS = {
    # BASELINE.json configs[1]: literal 2^20 x 256 B (SURVEY.md 8(d) "C2"); gadget set of CFG_20_256 (util.rs:7-20)
    "c2": {"n": 2, "nu_1": 9, "nu_2": 11, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 1, "db_item_size": 256},
    # configs[0]: 2^14 x 256 B
    "c1": {"n": 2, "nu_1": 9, "nu_2": 5, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 1, "db_item_size": 256},
    # the reference's own packed preset CFG_20_256 (2^20 x 256 B packed into 2^15 x 8 KiB)
    "p2": {"n": 2, "nu_1": 9, "nu_2": 6, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 1, "db_item_size": 8192},
    # configs[3]: 2^20 items x 32 KiB (SpiralWiki-style payload): 16 planes, 256 GiB encoded (224 GiB packed)
    "c4": {"n": 2, "nu_1": 9, "nu_2": 11, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 4, "db_item_size": 32768},
    # configs[2] per-GPU view: 2^22 items x 256 B (nu = (9,13)); run with --gpus 8 (32 GiB of rows per GPU)
    "c3": {"n": 2, "nu_1": 9, "nu_2": 13, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
           "t_exp_right": 56, "instances": 1, "db_item_size": 256},
    "fast": {"n": 2, "nu_1": 6, "nu_2": 2, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
             "t_exp_right": 8, "instances": 1, "db_item_size": 8192},
}
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
Q = 268369921 * 249561089


def synthetic_wire_bytes(n_bytes, seed):
    """A syntactically valid serialized pp / query: 32-byte seed then LE u64 words < Q.  The answer path's
    arithmetic is data-independent, so throughput on these equals throughput on real ciphertexts; parity on
    real ciphertexts is what tests/ checks."""
    rng = np.random.default_rng(seed)
    body = rng.integers(0, Q, (n_bytes - 32) // 8, dtype=np.uint64)
    return rng.integers(0, 256, 32, dtype=np.uint8).tobytes() + body.tobytes()


def sweep_algorithmic_bytes(cfg, shards):
    """BASELINE.md section 2 / SURVEY.md 8(d): per query, per GPU shard (reference 8-byte word format)."""
    N = 2048
    T = cfg["instances"] * cfg["n"] ** 2
    dim0, num_per = 1 << cfg["nu_1"], 1 << cfg["nu_2"]
    return T * N * num_per * (dim0 // shards) * 8 + T * N * (dim0 // shards) * 2 * 8 + T * num_per * 4 * N * 8


def pmc_traffic(cfg_name, world, launches):
    """HBM bytes per sweep launch from the rocprofv3 PMC passes kept under profiles/ (FETCH_SIZE doubled per the
    gfx950 correction + WRITE_SIZE); bench.py cannot collect counters itself.  None when no matching record."""
    if cfg_name != "c2" or world != 1:
        return None
    path = os.path.join(ROOT, "profiles", "r01_pmc_sweep_c2_packed.json")
    try:
        rec = json.load(open(path))
        return rec["hbm_bytes_per_launch"] * rec.get("launches_per_query", 1) / launches
    except Exception:
        return None


def cpu_baseline(cfg_name, cfg):
    """CPU restatement of the reference (oracle/, kind "port") timed on this host, bounded sample."""
    import oracle
    o = oracle.Params(cfg)
    threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    cl = oracle.Client(o)
    pp = cl.generate_keys(11)
    q = cl.generate_query(12345 % o.num_items, 12)
    N, dim0, num_per, planes = 2048, o.dim0, o.num_per, o.instances * o.n * o.n
    t0 = time.time()
    v_reg, v_fold = o.expand_query(pp, q)
    v_neg = o.get_v_folding_neg(v_fold)
    t_expand = time.time() - t0
    # sweep: nz z-rows of one plane on random words (the u128 MAC loop is data independent)
    rng = np.random.default_rng(5)
    nz = max(1, min(N, (1 << 25) // (num_per * dim0)))
    dbs = rng.integers(0, 1 << 28, nz * num_per * dim0, dtype=np.uint64) * np.uint64((1 << 32) + 1) % np.uint64(1 << 60)
    t0 = time.time()
    oracle.sweep_rows(dbs, v_reg[:nz * dim0 * 2], nz, dim0, num_per)
    t_sweep = (time.time() - t0) * (N / nz) * planes
    # fold: a 2^k-leaf subtree (from_ntt of the leaves + 2^k - 1 fold steps), scaled by step count
    k = min(o.db_dim_2, 5)
    leaves = 1 << k
    cts_ntt = np.concatenate([rng.integers(0, 249561089, leaves * 2 * 2 * N, dtype=np.uint64)])
    t0 = time.time()
    raw = o.from_ntt(cts_ntt)
    if k > 0:
        o.fold_ciphertexts(raw, v_fold[:k * 2 * 2 * o.t_gsw * 2 * N], v_neg[:k * 2 * 2 * o.t_gsw * 2 * N], nu=k)
    t_sub = time.time() - t0
    t_fold = t_sub * (num_per / leaves) * planes
    total = t_expand + t_sweep + t_fold
    return {
        "value": 1.0 / total, "unit": "queries/s", "cores": threads, "kind": "port",
        "sample": ("C++ restatement of spiral-rs (oracle/), config %s: expand_query+get_v_folding_neg in full (%.2f s, "
                   "%d OpenMP threads = the reference's rayon loops); multiply_reg_by_database timed on %d of %d z-rows "
                   "of one plane and scaled x%d planes (%.1f s/query, single thread as server.rs:682-694); "
                   "from_ntt+fold_ciphertexts timed on a %d-leaf subtree and scaled to %d leaves x %d planes (%.1f s/query, "
                   "single thread); pack/encode omitted (<1%%)" %
                   (cfg_name, t_expand, threads, nz, N, planes, t_sweep, leaves, num_per, planes, t_fold)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=os.environ.get("SPIRAL_BENCH_CONFIG", "c2"), choices=sorted(CONFIGS))
    ap.add_argument("--sweep-iters", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=1,
                    help="queries per step (N = 1 only); > 1 uses sp_process_query_batch: groups of <= 8 queries "
                         "share one database pass (BASELINE configs[4]).  Default 1 = the single-query metric.")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    import torch
    import torch.distributed as dist
    import sdk_amd as sp
    from sdk_amd.sharding import gather_local, local_cts_tensor, partial_tensor, reduce_partials, scatter_fold_query

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local_rank)
    if sp.lib().sp_set_device(local_rank) != 0:
        raise SystemExit("sp_set_device failed")
    # SPIRAL_FORCE_DIST=1: run the N > 1 code path (process group, collectives) at world size 1 (1-GPU boxes)
    use_dist = world > 1 or os.environ.get("SPIRAL_FORCE_DIST") == "1"
    real_stdout = None
    if use_dist:
        # RCCL prints a version banner through C stdio on stdout; keep stdout for the one JSON line
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    cfg = CONFIGS[args.config]
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, synthetic_wire_bytes(p.setup_bytes(), 1))
    queries = [synthetic_wire_bytes(p.query_bytes(), 100 + i) for i in range(4)]
    mode = os.environ.get("SPIRAL_MULTIGPU", "scatter") if use_dist else "single"
    if mode in ("scatter", "columns") and (1 << cfg["nu_2"]) < world:
        mode = "reduce"
    db = sp.Database(p, rank, world, by_columns=(mode == "columns")).fill_synthetic(0x123456789)  # util.rs:171-173 seed
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    distributed_fold = mode == "scatter"
    # stream-ordered flow with the reduce-scatter of plane p overlapping the sweep of plane p+1; verified below
    # against the host-synchronised flow of the same data path before it is timed
    overlap = os.environ.get("SPIRAL_OVERLAP", "1") != "0"

    def step(i):
        if args.batch > 1 and world == 1:
            outs = sp.process_query_batch(p, pp, [queries[(i + k) % len(queries)] for k in range(args.batch)], db)
            return outs[0], None
        run = sp.QueryRun(p, pp, queries[i % len(queries)], db=db)  # row shards: expansion pruned to the shard's rows
        if mode == "columns":
            # column shards: complete outputs per shard, no partial sums; only the folded cts are gathered
            run.sweep(db)
            run.fold_local(run.partial_ptr(), world)
            run.sync()
            gathered = gather_local(local_cts_tensor(run), rank, world, dst=0)
            torch.cuda.synchronize()
            out = run.finish_gathered(gathered.data_ptr(), world) if rank == 0 else None
        elif distributed_fold:
            # row-sharded sweep -> RCCL reduce-scatter over columns -> every rank folds its columns ->
            # gather of one ciphertext per plane per rank -> rank 0 folds the last log2(N) levels
            out = scatter_fold_query(run, db, rank, world, overlap=overlap)
        elif use_dist:
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

    if distributed_fold and overlap:
        # self-check: the overlapped flow must reproduce the synchronised flow's response byte for byte
        overlap = False
        ref, _ = step(0)
        overlap = True
        try:
            got, _ = step(0)
            good = rank != 0 or got == ref
        except Exception as e:  # an API error on any rank sends every rank back to the synchronised flow
            print("bench: overlapped flow raised %r" % (e,), file=sys.stderr, flush=True)
            good = False
        ok = torch.tensor([1 if good else 0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            overlap = False
            if rank == 0:
                print("bench: overlapped multi-GPU flow disagreed with the synchronised flow; timing the latter",
                      file=sys.stderr, flush=True)
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    stage = np.zeros(4)
    for i in range(args.steps):
        out, t = step(i)
        if t is not None:
            stage += np.array(t)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed * 1e3 / args.steps

    # dominant kernel: the db sweep.  HIP events on the launch stream around `sweep_iters` launches.
    run = sp.QueryRun(p, pp, queries[0], db=db)
    if distributed_fold and overlap:                          # one launch per plane, exchange overlapped
        sweep_ms, launches = run.bench_sweep(db, args.sweep_iters, per_plane=1), cfg["instances"] * cfg["n"] ** 2
    else:                                                     # 1, or one per plane when the fold is overlapped
        sweep_ms, launches = run.bench_sweep(db, args.sweep_iters), run.sweep_launches(db)
    run.free()
    alg_bytes = sweep_algorithmic_bytes(cfg, world) / launches
    standalone = alg_bytes / (sweep_ms * 1e-3) / 1e9
    # In the timed region the sweep launches are the only work on the query's main stream between the "expanded" and
    # "swept" HIP events, so that span / launches is the kernel's average duration in the real run -- including what
    # it loses to the folds that share the CUs from the second stream.  (Batched steps have no per-query span.)
    in_situ_ms = stage[1] / args.steps / launches if (stage[1] > 0 and args.batch == 1) else sweep_ms
    achieved = alg_bytes / (in_situ_ms * 1e-3) / 1e9

    if rank == 0:
        line = {
            "metric": "PIR queries/sec (single query, full answer path) on 2^%d items x %d B" %
                      (cfg["nu_1"] + cfg["nu_2"], cfg["db_item_size"]),
            "value": args.steps * (args.batch if world == 1 else 1) / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"arithmetic": "exact integers: u32 x u32 -> u64 multiply-accumulate mod two 28-bit primes, u32 NTT "
                                     "butterflies (Shoup), no floating point",
                       "workload": "spiral-rs process_query, %s = %s, encoded DB %.1f GiB resident in HBM, "
                                   "%s" % (args.config, json.dumps(cfg, sort_keys=True),
                                           p.db_words * 8 / 2**30,
                                           "unsharded" if world == 1 else {"scatter": "row-sharded dim0/%d per GPU + RCCL reduce-scatter of partial Regev cts%s, distributed fold, all-gather" % (world, " (per plane, overlapping the next plane's sweep)" if overlap else ""), "reduce": "row-sharded dim0/%d per GPU + RCCL reduce onto rank 0" % world, "columns": "column-sharded num_per/%d per GPU, distributed fold, all-gather (no partial sums)" % world}[mode]),
                       "queries_per_step": args.batch if world == 1 else 1,
                       "stage_ms": {"expand": stage[0] / args.steps, "sweep": stage[1] / args.steps,
                                    "fold": stage[2] / args.steps, "pack_encode": stage[3] / args.steps}},
            "roofline": {"bound": "hbm", "kernel": "k_sweep_packed_persist<4>" if cfg["nu_2"] >= 7 else "k_sweep_narrow2",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(args.config, world, launches),
                         "algorithmic_bytes_per_launch": alg_bytes, "ms_per_launch": in_situ_ms,
                         "launches_per_query": launches,
                         "standalone": {"ms_per_launch": sweep_ms, "achieved": standalone, "frac": standalone / HBM_PEAK_GBPS,
                                        "note": "the same launches issued back to back with nothing else on the GPU "
                                                "(sp_bench_sweep, HIP events on the launch stream)"},
                         "note": "achieved = algorithmic bytes (reference 8-byte words) / average duration of a sweep "
                                 "launch inside the timed steps (HIP events on its stream; the fold of the previous plane "
                                 "runs beside it); the resident database is bit-packed to 7 bytes per word, so HBM traffic "
                                 "(PMC, profiles/r01_pmc_sweep_c2_packed.json) is below the algorithmic bytes"},
        }
        if world == 1 and not args.no_cpu_baseline:
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
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
