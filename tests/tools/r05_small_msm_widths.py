#!/usr/bin/env python3
"""Round 5: STANDALONE latency (one call + synchronise, median of 11, reduce phase in line) and phase times of a single MSM of 2^lg terms over a
2^lg-point SRS for the narrow window widths -- what a small proof's round-ending commitment costs.  python tests/tools/r05_small_msm_widths.py [lg ...]"""
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
bbg.set_option("msm_async_reduce", 0)
for lg in [int(a) for a in sys.argv[1:]] or [14, 15, 16, 17]:
    n = 1 << lg
    srs = bbg.srs_synth_hashed(0xBB254, n)
    d_sc = torch.from_numpy(pkg.synthetic_scalars(7, n).view(np.int64).reshape(-1)).cuda()
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    for w in (13, 16, 17):
        bbg.set_option("msm_window", w)
        for _ in range(3):
            bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
        bbg.sync()
        ts = []
        for _ in range(11):
            t0 = time.perf_counter()
            bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
            bbg.sync()
            ts.append(time.perf_counter() - t0)
        bbg.profile_enable(True)
        for _ in range(10):
            bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
        bbg.sync()
        pr = {k: bbg.profile_get(k) for k in ("msm_recode", "msm_sort", "msm_accumulate", "msm_reduce")}
        bbg.profile_enable(False)
        print("2^%d C=%d standalone %.3f ms  phases ms %s" % (lg, w, sorted(ts)[5] * 1e3, {k: round(v[0] / 10, 4) for k, v in pr.items()}), flush=True)
    bbg.set_option("msm_window", 0)
    srs.free()
