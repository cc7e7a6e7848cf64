"""Helper for test_gpu_parity.test_rccl_world1_paths: the three N > 1 bench paths driven through a REAL
nccl (= RCCL) process group of world size 1 on cuda:0 — the collectives run on the library's own
hipMalloc'd buffers (zero-copy int32/int64 views), so dtype / pointer / stream handling is the same code
the 8-GPU bench uses.  Prints 'rccl-ok' when all responses equal the oracle's."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
import torch
import torch.distributed as dist

import oracle as oracle_mod
import sdk_amd as sp
from sdk_amd.sharding import (gather_local, local_cts_tensor, partial_tensor, reduce_partials,
                              reduce_scatter_partials)

CFG = {"n": 2, "nu_1": 6, "nu_2": 7, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
       "t_exp_right": 8, "instances": 1, "db_item_size": 256}


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    o = oracle_mod.Params(CFG)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(5)
    idx = 4321 % o.num_items
    q = cl.generate_query(idx, 6)
    item, db = o.generate_random_db_and_get_item(idx)
    expect = o.process_query(pp, q, db)
    p = sp.Params(CFG)
    gpp = sp.PublicParameters.deserialize(p, pp)
    rank, world = 0, 1

    # scatter: sweep_scatter -> reduce_scatter_tensor -> fold_local -> all_gather_into_tensor -> finish_gathered
    gdb = sp.Database(p, rank, world).load(db)
    run = sp.QueryRun(p, gpp, q)
    run.sweep_scatter(gdb, world)
    run.sync()
    mine = reduce_scatter_partials(partial_tensor(run), rank, world)
    torch.cuda.synchronize()
    run.fold_local(mine.data_ptr(), world)
    run.sync()
    gathered = gather_local(local_cts_tensor(run), rank, world, dst=0)
    torch.cuda.synchronize()
    assert run.finish_gathered(gathered.data_ptr(), world) == expect, "scatter path"
    run.free()

    # the same path as bench.py drives it: stream-ordered with per-plane async reduce-scatter, and host-synchronised
    from sdk_amd.sharding import scatter_fold_query
    for overlap, per_plane in ((True, False), (False, False), (True, True), (True, False)):
        run = sp.QueryRun(p, gpp, q, db=gdb)
        assert scatter_fold_query(run, gdb, rank, world, overlap=overlap, fold_per_plane=per_plane) == expect, \
            "scatter_fold_query overlap=%s fold_per_plane=%s" % (overlap, per_plane)
        run.free()

    # the same flow with the collectives issued by the library (sp_process_query_sharded, RCCL linked directly):
    # a real ncclCommInitRank / ncclReduceScatter / ncclAllGather on this process's GPU
    from sdk_amd.sharding import Comm
    comm = Comm.rccl(0, 1, Comm.unique_id())
    sp.paths_taken()
    for _ in range(2):
        assert comm.process_query(p, gpp, q, gdb) == expect, "sp_process_query_sharded over RCCL"
    assert "rccl_in_library" in sp.paths_taken()
    comm.barrier()
    assert len(comm.timings()) == 3
    info = comm.describe()      # sp_comm_describe: the real RCCL's version, the sizes of this query's collectives
    assert info["transport"] == "rccl" and info["rccl_version"] > 20000 and info["world"] == 1, info
    assert info["reduce_scatter_u32"]["per_plane_recv_bytes"] > 0 and info["last_query_ms"]["sweeps_with_overlapped_exchange"] > 0, info
    print("comm:", info)
    comm.free()

    # reduce: sweep -> dist.reduce -> finish
    run = sp.QueryRun(p, gpp, q).sweep(gdb)
    run.sync()
    reduce_partials(partial_tensor(run), dst=0)
    torch.cuda.synchronize()
    assert run.finish() == expect, "reduce path"
    run.free()

    # columns: sweep of a column shard -> fold_local -> gather -> finish_gathered
    cdb = sp.Database(p, rank, world, by_columns=True).load(db)
    run = sp.QueryRun(p, gpp, q).sweep(cdb)
    run.fold_local(run.partial_ptr(), world)
    run.sync()
    gathered = gather_local(local_cts_tensor(run), rank, world, dst=0)
    torch.cuda.synchronize()
    assert run.finish_gathered(gathered.data_ptr(), world) == expect, "columns path"
    run.free()
    assert cl.decode_response(expect) == o.item_to_vec(item)
    dist.barrier()
    dist.destroy_process_group()
    print("rccl-ok")


if __name__ == "__main__":
    main()
