#!/usr/bin/env python3
"""What the per-scope HIP events of bbg_profile_enable cost the bench step (they sit between the kernels of the main stream)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
lg = 20; n = 1 << lg
bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
bbg.set_option("msm_async_reduce", 1)
srs = bbg.srs_synth_hashed(0xBB254, n)
d_sc = torch.from_numpy(pkg.synthetic_scalars(0xBB254 + 3, n).view(np.int64)).cuda()
d_c = torch.from_numpy(pkg.synthetic_scalars(0xBB254 + 120, n).view(np.int64)).cuda()
out = torch.zeros(12, dtype=torch.int64, device="cuda")
bbg.ntt_prepare(lg)
def step():
    bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
    bbg.ntt_device(d_c.data_ptr(), lg, 0)
def run(profile):
    for _ in range(5): step()
    bbg.join(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        bbg.profile_enable(profile)
        t0 = time.perf_counter()
        for _ in range(20): step()
        bbg.join(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
        if profile:
            for k in ("msm_recode", "msm_sort", "msm_offsets", "msm_accumulate", "msm_reduce", "ntt_pass"): bbg.profile_get(k)
        bbg.profile_enable(False)
    return best
for p in (False, True, False, True):
    print(f"profile events {'on ' if p else 'off'}: {run(p):.4f} ms/step", flush=True)
