"""Per-rank compute time of the N > 1 path (row shard + distributed fold), measured on ONE GPU: rank 0's share of
config C2 for G = 1, 2, 4, 8 with the collectives replaced by local stand-ins (no xGMI traffic).  Gives the
compute floor of the scaling curve; the difference to the driver's N-GPU numbers is exchange + sync time."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
import sdk_amd as sp
from sdk_amd.sharding import local_cts_tensor, partial_tensor


def main():
    cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    q = bench.synthetic_wire_bytes(p.query_bytes(), 100)
    for G in (8, 4, 2):
        db = sp.Database(p, 0, G).fill_synthetic(0x123456789)
        torch.cuda.synchronize()
        acc = [0.0] * 5
        iters = 6
        for it in range(iters + 2):
            t = [time.perf_counter()]
            run = sp.QueryRun(p, pp, q, db=db if os.environ.get("NO_PRUNE") != "1" else None)
            run.sync(); t.append(time.perf_counter())
            run.sweep_scatter(db, G)
            run.sync(); t.append(time.perf_counter())
            part = partial_tensor(run)
            mine = part[: part.numel() // G]
            run.fold_local(mine.data_ptr(), G)
            run.sync(); t.append(time.perf_counter())
            gathered = local_cts_tensor(run).repeat(G).contiguous()
            torch.cuda.synchronize(); t.append(time.perf_counter())
            run.finish_gathered(gathered.data_ptr(), G)
            t.append(time.perf_counter())
            run.free()
            if it >= 2:
                for k in range(5):
                    acc[k] += (t[k + 1] - t[k]) * 1e3 / iters
        print("G=%d  begin %.3f  sweep %.3f  fold_local %.3f  (gather stand-in %.3f)  finish %.3f  total %.3f ms" %
              (G, *acc, sum(acc)), flush=True)
        del db
        torch.cuda.synchronize()


if __name__ == "__main__" and not (len(sys.argv) > 2 and sys.argv[2] == "flow"):
    main()


def flow_emulation(cfg_name="c2", iters=8, per_plane_fold=True):
    """The overlapped N > 1 flow of sdk_amd.sharding.scatter_fold_query on ONE GPU, rank 0's shard, with the
    per-plane reduce-scatter replaced by a device copy of the rank's own chunk on a third stream (no xGMI traffic,
    no remote partial sums): the stream structure, launch counts and overlaps are the real ones."""
    cfg = bench.CONFIGS[cfg_name]
    p = sp.Params(cfg)
    pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
    q = bench.synthetic_wire_bytes(p.query_bytes(), 100)
    planes = cfg["instances"] * cfg["n"] ** 2
    comm = torch.cuda.Stream()
    for G in (8, 4, 2):
        db = sp.Database(p, 0, G).fill_synthetic(0x123456789)
        torch.cuda.synchronize()
        tot = 0.0
        for it in range(iters + 2):
            t0 = time.perf_counter()
            run = sp.QueryRun(p, pp, q, db=db)
            main, second = torch.cuda.ExternalStream(run.stream()), torch.cuda.ExternalStream(run.stream2())
            part = partial_tensor(run)
            pw = part.numel() // planes
            chunk = pw // G
            with torch.cuda.stream(main):
                mine = torch.empty(planes * chunk, dtype=part.dtype, device=part.device)
            for pl in range(planes):
                run.sweep_scatter_plane(db, G, pl)
                comm.wait_stream(main)
                with torch.cuda.stream(comm):
                    mine[pl * chunk:(pl + 1) * chunk].copy_(part[pl * pw:pl * pw + chunk])
                if per_plane_fold:
                    second.wait_stream(comm)
                    run.fold_local_plane(mine[pl * chunk:].data_ptr(), G, pl)
            if per_plane_fold:
                run.fold_local_join()
            else:
                main.wait_stream(comm)
                run.fold_local(mine.data_ptr(), G)
            with torch.cuda.stream(main):
                gathered = local_cts_tensor(run).repeat(G).contiguous()
            run.finish_gathered(gathered.data_ptr(), G)
            run.sync()
            run.free()
            if it >= 2:
                tot += (time.perf_counter() - t0) * 1e3 / iters
        print("flow G=%d  per-plane fold %s  %.3f ms per query (rank-local work of the overlapped flow, exchange replaced by a local copy)" % (G, per_plane_fold, tot),
              flush=True)
        del db
        torch.cuda.synchronize()


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "flow":
    flow_emulation(sys.argv[1], per_plane_fold=True)
    flow_emulation(sys.argv[1], per_plane_fold=False)
