"""One rank's critical path of a G-rank row-sharded query, measured alone on one GPU: shard 0 of G of
the database, the real per-plane sweeps / local fold / final levels, and a transport whose two collectives return at once
(sdk_amd.sharding.NullTransport) -- i.e. everything a rank does except waiting for its peers' data.  The exchange time
itself is taken from the payloads and the xGMI link rate.  Two figures per configuration:
  serial    : sp_process_query_sharded per query (expansion on the critical path)
  pipelined : sp_process_queries_sharded over a list (query k + 1 expands while query k is swept)
Results are timing only (the null transport moves no data).  Usage: python scripts/r05/rank_critical_path.py [c2 c3] [G]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import sdk_amd as sp  # noqa: E402
from conftest import C1, C2  # noqa: E402
from sdk_amd.sharding import NullTransport  # noqa: E402

CFG = {"c1": C1, "c2": C2, "c3": dict(C2, nu_2=13)}
XGMI_GBPS = 153.0   # per link and direction


def run(name, G, n_q=12):
    cfg = CFG[name]
    o = oracle.Params(cfg)
    cl = oracle.Client(o)
    pp = cl.generate_keys(7)
    qs = [cl.generate_query((7919 * k + 1) % o.num_items, 100 + k) for k in range(n_q)]
    p = sp.Params(cfg)
    gpp = sp.PublicParameters.deserialize(p, pp)
    shard = sp.Database(p, 0, G).fill_synthetic(0x123456789)
    comm = NullTransport(0, G).comm
    comm.reserve(p)
    for q in qs[:2]:
        comm.process_query(p, gpp, q, shard)
    sp.lib().sp_set_device(0)
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tm = [0.0, 0.0]
    for q in qs:
        comm.process_query(p, gpp, q, shard)
        t = comm.timings()
        tm[0] += t[0]
        tm[1] += t[1]
    torch.cuda.synchronize()
    serial_ms = (time.perf_counter() - t0) * 1e3 / n_q
    comm.process_queries(p, gpp, qs[:3], shard)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    comm.process_queries(p, gpp, qs, shard)
    torch.cuda.synchronize()
    pipe_ms = (time.perf_counter() - t0) * 1e3 / n_q
    planes = o.instances * o.n * o.n
    rs_bytes_per_link = planes * 4 * 2048 * o.num_per * 4 / G          # each rank sends 1/G of its partial buffer to every peer
    ag_bytes = planes * 2 * 2048 * 8                                    # one raw ciphertext per plane per rank
    xchg_ms = rs_bytes_per_link / (XGMI_GBPS * 1e9) * 1e3
    plane_sweep_ms = tm[0] / n_q / planes
    exposed_rs_ms = max(0.0, xchg_ms / planes - 0.0)                    # the last plane's reduce-scatter cannot hide
    out = {"config": name, "G": G, "shard_resident_GiB": shard.device_bytes() / 2**30,
           "serial_ms_per_query": serial_ms, "pipelined_ms_per_query": pipe_ms,
           "sweep_span_ms": tm[0] / n_q, "tail_fold_gather_ms": tm[1] / n_q, "sweep_ms_per_plane": plane_sweep_ms,
           "reduce_scatter_MiB_per_link": rs_bytes_per_link / 2**20, "reduce_scatter_ms_at_153GBps": xchg_ms,
           "exposed_exchange_ms_estimate": exposed_rs_ms + ag_bytes * G / (XGMI_GBPS * 1e9) * 1e3,
           "implied_qps_serial": 1e3 / (serial_ms + exposed_rs_ms), "implied_qps_pipelined": 1e3 / (pipe_ms + exposed_rs_ms),
           "note": "one rank alone on one MI355X with a null transport: what the rank computes, not a measured %d-GPU run; the "
                   "exchange of plane p overlaps the sweep of plane p + 1 when %.2f ms (per plane, one link) <= %.2f ms (sweep "
                   "of a plane); the last plane's exchange is added as exposed" % (G, xchg_ms / planes, plane_sweep_ms)}
    comm.free()
    return out


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in CFG] or ["c2"]
    Gs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8]
    for n in names:
        for G in Gs:
            print(json.dumps(run(n, G)), flush=True)
