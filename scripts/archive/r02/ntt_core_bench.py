import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdk_amd as sp
p = sp.Params({"n": 2, "nu_1": 6, "nu_2": 2, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8, "t_exp_right": 8})
L = sp.lib()
for M in (1, 2, 4):
    for blocks in (256, 512, 1024, 2048, 4096):
        ns = C.c_float(0)
        rc = L.sp_bench_ntt(C.c_void_p(p.h), C.c_int(M), C.c_int(blocks), C.c_int(64), C.byref(ns))
        print(f"M={M} blocks={blocks:5d} ({blocks/256:4.1f}/CU)  {ns.value:6.2f} ns per NTT" if rc == 0 else L.sp_last_error())
