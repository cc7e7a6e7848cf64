"""Interference probe: the C2 db sweep timed alone and while a pure-ALU co-runner (the NTT core micro-benchmark,
no memory traffic but twiddles) occupies the other wave slots from a second stream."""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench
import sdk_amd as sp

cfg = bench.CONFIGS["c2"]
p = sp.Params(cfg)
pp = sp.PublicParameters.deserialize(p, bench.synthetic_wire_bytes(p.setup_bytes(), 1))
q = bench.synthetic_wire_bytes(p.query_bytes(), 100)
db = sp.Database(p).fill_synthetic(0x123456789)
run = sp.QueryRun(p, pp, q)
L = sp.lib()
print("alone: %.3f ms" % run.bench_sweep(db, 3))
print("alone: %.3f ms" % run.bench_sweep(db, 3))
for M, blocks in ((2, 256), (2, 512), (4, 256), (1, 256), (1, 1024)):
    ns = C.c_float(0)
    reps = int(120e6 / (blocks * (M if M < 4 else 4) * 5.0))

    def co():
        L.sp_bench_ntt(C.c_void_p(p.h), C.c_int(M), C.c_int(blocks), C.c_int(reps), C.byref(ns))

    th = threading.Thread(target=co)
    t0 = time.perf_counter()
    th.start()
    time.sleep(0.02)
    ms = run.bench_sweep(db, 3)
    t1 = time.perf_counter()
    th.join()
    t2 = time.perf_counter()
    print("co-run M=%d blocks=%d: sweep %.3f ms  (co-runner %.2f ns/ntt; sweep call ended %.0f ms, co-runner %.0f ms after start)" %
          (M, blocks, ms, ns.value, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
