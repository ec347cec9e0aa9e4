#!/usr/bin/env python3
"""Workload for the profiler: config 5's single-GPU legs -- three 2^24-point MSMs (hashed SRS, C = 22 window tables: 12.9 GiB, far beyond
the Infinity Cache) and three 2^24 coset NTTs, device resident; prints the wall-clock per leg."""
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
bbg.set_option("msm_async_reduce", 1)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << lg
srs = bbg.srs_synth_hashed(0xBB254, n)
d_sc = torch.from_numpy(pkg.synthetic_scalars(0xBB254 + 24, n).view(np.int64).reshape(-1)).cuda()
d_x = torch.from_numpy(pkg.synthetic_scalars(900 + lg, n).view(np.int64).reshape(-1)).cuda()
out = torch.zeros(12, dtype=torch.int64, device="cuda")
bbg.ntt_prepare(lg)
bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr()); bbg.join(); bbg.sync()
bbg.ntt_device(d_x.data_ptr(), lg, 2); bbg.sync()
tm, tn = [], []
for _ in range(3):
    t0 = time.perf_counter()
    bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr()); bbg.join(); bbg.sync()
    tm.append(time.perf_counter() - t0)
for _ in range(3):
    t0 = time.perf_counter()
    bbg.ntt_device(d_x.data_ptr(), lg, 2); bbg.sync()
    tn.append(time.perf_counter() - t0)
print(f"legs 2^{lg}: msm_ms {min(tm)*1e3:.3f} ntt_ms {min(tn)*1e3:.3f}")
