#!/usr/bin/env python3
"""MSM latency at small n: the 12-bit window configuration against the 16-bit one (device-resident scalars, stand-alone = one call
+ sync, pipelined = 20 calls back to back with the reduce phase on the auxiliary stream).  Also checks both against the oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

pkg = ge.load_package()
from oracle.oracle import Oracle  # noqa: E402

O = Oracle()
bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
N = 1 << 18
srs = bbg.srs_synth_hashed(0xBB254, N)
sc = pkg.synthetic_scalars(7, N)
d_sc = torch.from_numpy(sc.view(np.int64).reshape(-1)).cuda()
out = torch.zeros(12, dtype=torch.int64, device="cuda")
pts = srs.read(0, 4096)
for w in (12, 16):
    bbg.set_option("msm_window", w)
    for n in (1, 3, 100, 4096):
        got = O.jac_to_affine(bbg.msm(srs, sc[:n]))
        assert np.array_equal(got, O.pippenger(sc[:n], pts[:n])), (w, n)
print("window 12 and 16 agree with the oracle at n = 1, 3, 100, 4096")
print("log2n  window  standalone_ms  pipelined_ms")
for lg in range(10, 19):
    n = 1 << lg
    for w in (12, 16):
        bbg.set_option("msm_window", w)
        for mode in ("standalone", "pipelined"):
            bbg.set_option("msm_async_reduce", 1 if mode == "pipelined" else 0)
            for _ in range(3):
                bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
            bbg.join(); bbg.sync()
            if mode == "standalone":
                ts = []
                for _ in range(20):
                    t0 = time.perf_counter()
                    bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
                    bbg.sync()
                    ts.append(time.perf_counter() - t0)
                sa = sorted(ts)[len(ts) // 2] * 1e3
            else:
                t0 = time.perf_counter()
                for _ in range(40):
                    bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
                bbg.join(); bbg.sync()
                pl = (time.perf_counter() - t0) / 40 * 1e3
        print(f"{lg:5d}  {w:6d}  {sa:13.3f}  {pl:12.3f}", flush=True)
