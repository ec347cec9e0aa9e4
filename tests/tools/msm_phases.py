#!/usr/bin/env python3
"""Per-phase times (HIP events) of the pipelined MSM at a given size for both window widths: python tests/tools/msm_phases.py [log2n ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

pkg = ge.load_package()
bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
sizes = [int(a) for a in sys.argv[1:]] or [20, 21]
N = 1 << max(sizes)
srs = bbg.srs_synth_hashed(0xBB254, N)
sc = pkg.synthetic_scalars(7, N)
d_sc = torch.from_numpy(sc.view(np.int64).reshape(-1)).cuda()
out = torch.zeros(12, dtype=torch.int64, device="cuda")
bbg.set_option("msm_async_reduce", 1)
for lg in sizes:
    n = 1 << lg
    for w in (13, 16, 17, 19, 20, 22):
        bbg.set_option("msm_window", w)
        for _ in range(3):
            bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
        bbg.join(); bbg.sync()
        bbg.profile_enable(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
        bbg.join()
        e1.record()
        bbg.sync()
        pr = {k: bbg.profile_get(k) for k in ("msm_recode", "msm_sort", "msm_accumulate", "msm_reduce")}
        bbg.profile_enable(False)
        print(lg, w, "per MSM %.3f ms" % (e0.elapsed_time(e1) / 20), {k: round(v[0] / max(1, v[1]), 4) for k, v in pr.items()}, flush=True)
