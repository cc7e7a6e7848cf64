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


if __name__ == "__main__":
    main()
