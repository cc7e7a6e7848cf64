"""One rank of tests/test_emulated_library.py::test_row_sharded_query_over_process_ranks (TEST INFRASTRUCTURE).

Run as N processes with SPIRAL_HIP_LIB = the emulated build: each loads its row shard of the database, joins the library's own
communicator (sp_comm_create -> comm.cpp -> the shared-memory stand-in for RCCL, tests/emu/emu_rccl.cpp) and answers the same
queries through sp_process_query_sharded / sp_process_queries_sharded; rank 0 compares with the oracle and prints 'sharded-ok'.
usage: _emu_sharded_rank.py RANK WORLD ID_FILE CONFIG_NAME"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import oracle as oracle_mod  # noqa: E402
import sdk_amd as sp  # noqa: E402
from conftest import FAST, FAST56  # noqa: E402
from sdk_amd.sharding import Comm  # noqa: E402

CONFIGS = {"narrow": dict(FAST56, nu_2=4), "packed": dict(FAST, nu_1=6, nu_2=7, db_item_size=256),
           "odd-gadgets": dict(FAST56, nu_2=5, t_gsw=3, t_conv=3, t_exp_left=5)}


def main():
    rank, world, id_file, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    assert hasattr(sp.lib(), "sp_emulated_device_marker"), "this helper is for the emulated build only"
    cfg = CONFIGS[name]
    o = oracle_mod.Params(cfg)
    cl = oracle_mod.Client(o)
    pp = cl.generate_keys(11)
    idx = 1234 % o.num_items
    q, q2 = cl.generate_query(idx, 12), cl.generate_query((idx * 7 + 3) % o.num_items, 13)
    item, db = o.generate_random_db_and_get_item(idx)
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    shard = sp.Database(p, rank, world).load(db)
    if rank == 0:
        ident = Comm.unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(ident)
        os.rename(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            assert time.time() - t0 < 120, "rank 0 never published the communicator id"
            time.sleep(0.01)
        ident = open(id_file, "rb").read()
    comm = Comm.rccl(rank, world, ident)
    comm.reserve(p)
    sp.paths_taken()
    outs = [comm.process_query(p, gpp, qq, shard) for qq in (q, q2, q)]
    taken = sp.paths_taken()
    assert "rccl_in_library" in taken
    if os.environ.get("SPIRAL_EXPAND_SPLIT") == "1":      # (the caller forces the split expansion of wide row shards on this shape)
        assert "expand_split" in taken and "expand_pruned" in taken, taken
    info = comm.describe()      # sp_comm_describe: what bench.py --gpus N logs per rank
    assert info["world"] == world and info["rank"] == rank and info["transport"] == "rccl" and info["planes"] == o.instances * o.n * o.n
    assert info["reduce_scatter_u32"]["per_plane_recv_bytes"] == 4 * 2048 * o.num_per // world * 4
    assert info["all_gather_u64"]["send_bytes"] == info["planes"] * 2 * 2048 * 8
    assert info["last_query_ms"]["exposed_exchange_after_last_sweep"] >= 0
    listed = comm.process_queries(p, gpp, [q, q2, q, q2], shard)
    comm.barrier()
    comm.free()
    if rank == 0:
        want = [o.process_query(pp, qq, db) for qq in (q, q2)]
        assert outs == [want[0], want[1], want[0]], "sp_process_query_sharded differs from the oracle"
        assert listed == [want[0], want[1], want[0], want[1]], "sp_process_queries_sharded differs from the oracle"
        if name != "odd-gadgets":   # (that gadget set is a shape test: its noise does not leave room to decode)
            assert cl.decode_response(outs[0]) == o.item_to_vec(item)
        print("sharded-ok")
    else:
        assert outs == [b"", b"", b""] and listed == []


if __name__ == "__main__":
    main()
