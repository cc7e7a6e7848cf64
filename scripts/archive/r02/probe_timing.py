import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
def T(msg):
    global t0
    t = time.time(); print(f"{t - t0:8.3f}s  {msg}", flush=True); t0 = t
import numpy as np
import oracle
T("import oracle")
import sdk_amd as sp
T("import sdk_amd")
cfg = {"n": 2, "nu_1": 6, "nu_2": 2, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8, "t_exp_right": 8, "instances": 1, "db_item_size": 8192}
o = oracle.Params(cfg); T("oracle.Params")
cl = oracle.Client(o); T("oracle.Client")
pp = cl.generate_keys(1); T("generate_keys")
q = cl.generate_query(5, 2); T("generate_query")
item, db = o.generate_random_db_and_get_item(5); T("gen db")
r_cpu = o.process_query(pp, q, db); T("oracle process_query")
print("devices", sp.lib().sp_device_count()); T("device count")
p = sp.Params(cfg); T("sp.Params")
gpp = sp.PublicParameters.deserialize(p, pp); T("pp deserialize")
gdb = sp.Database(p); T("db create")
gdb.load(db); T("db load")
r = sp.process_query(p, gpp, q, gdb); T("process_query 1")
r = sp.process_query(p, gpp, q, gdb); T("process_query 2")
print("match", r == r_cpu)
x = np.zeros(4096, dtype=np.uint64); x[0] = 100; x[2048] = 100
y = sp.ntt_forward(p, x); T("ntt_forward"); print((y == 100).all())
y = sp.to_ntt(p, x[:2048]); T("to_ntt")
p2 = sp.Params(cfg); T("sp.Params 2")
y = sp.to_ntt(p2, x[:2048]); T("to_ntt on new params")
