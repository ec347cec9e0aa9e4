import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
bbg = pkg.Bbg(0)
for lg in (12, 16):
    c = pkg.synthetic_scalars(5, 1 << lg)
    for op in range(8):
        k = c[0] if op >= 4 else None
        t = time.time(); bbg.ntt(c, op, 0, k); t1 = time.time() - t
        t = time.time(); bbg.ntt(c, op, 0, k); t2 = time.time() - t
        print(lg, op, round(t1, 4), round(t2, 4), flush=True)
t = time.time(); bbg.set_option("ntt_max_logr8", 9); print("set_option", time.time() - t)
t = time.time(); bbg.ntt(c, 0); print("after plan change", time.time() - t)
from oracle.oracle import Oracle
O = Oracle()
t = time.time(); O.ntt(c, 0); print("oracle ntt 2^16", time.time() - t)
t = time.time(); O.canon(0, c); print("oracle canon", time.time() - t)
