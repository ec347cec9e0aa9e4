#!/usr/bin/env python3
"""Where the 20-bit window configuration (13 windows, 2^19 buckets) overtakes the 16-bit one (16 windows, 2^15 buckets):
stand-alone (call + sync) and pipelined (40 calls back to back, reduce phase on the auxiliary stream) MSM times per n."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

pkg = ge.load_package()
bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
LO, HI = 17, 22
N = 1 << HI
srs = bbg.srs_synth_hashed(0xBB254, N)
sc = pkg.synthetic_scalars(7, N)
d_sc = torch.from_numpy(sc.view(np.int64).reshape(-1)).cuda()
out = torch.zeros(12, dtype=torch.int64, device="cuda")
print("log2n  window  standalone_ms  pipelined_ms  result")
for lg in range(LO, HI + 1):
    n = 1 << lg
    res = {}
    for w in (16, 20):
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
                res[w] = out.cpu().numpy().copy()
            else:
                t0 = time.perf_counter()
                for _ in range(40):
                    bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
                bbg.join(); bbg.sync()
                pl = (time.perf_counter() - t0) / 40 * 1e3
        same = "" if w == 16 else ("same point" if np.array_equal(pkg.jac_equal_key(res[16]), pkg.jac_equal_key(res[20])) else "DIFFERENT") if hasattr(pkg, "jac_equal_key") else ""
        print(f"{lg:5d}  {w:6d}  {sa:13.3f}  {pl:12.3f}  {same}", flush=True)
