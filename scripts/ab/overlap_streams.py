#!/usr/bin/env python3
"""A/B: the bench step (1 MSM + 1 NTT at 2^20) with the NTT on a stream of its own, and the MSM's main stream at lower priority than
the reduce streams / the NTT stream.  python scripts/ab/overlap_streams.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
lg = 20; n = 1 << lg
def run(mode, reduce_prio):
    bbg = pkg.Bbg(0)
    hi, lo = -1, 0
    s_main = torch.cuda.Stream(priority=lo if mode == 2 else 0) if mode != 0 else torch.cuda.current_stream()
    s_ntt = torch.cuda.Stream(priority=hi if mode == 2 else 0)
    bbg.set_stream(s_main.cuda_stream)
    bbg.set_option("msm_reduce_priority", reduce_prio)
    bbg.set_option("msm_async_reduce", 1)
    srs = bbg.srs_synth_hashed(0xBB254, n)
    d_sc = torch.from_numpy(pkg.synthetic_scalars(0xBB254 + 3, n).view(np.int64)).cuda()
    d_c = torch.from_numpy(pkg.synthetic_scalars(0xBB254 + 120, n).view(np.int64)).cuda()
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    bbg.ntt_prepare(lg)
    def step():
        bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
        if mode in (1, 2):
            bbg.set_stream(s_ntt.cuda_stream)
        bbg.ntt_device(d_c.data_ptr(), lg, 0)
        if mode in (1, 2):
            bbg.set_stream(s_main.cuda_stream)
    for _ in range(5): step()
    bbg.join(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20): step()
        bbg.join(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    ref = out.cpu().numpy().copy()
    bbg.close()
    return best, ref
res = {}
for mode, name in ((0, "default stream (bench.py)"), (3, "one created stream"), (1, "NTT on its own stream"), (2, "NTT stream high, MSM main stream low")):
    for rp in (1, 0):
        ms, ref = run(mode, rp)
        print(f"{name:40s} msm_reduce_priority={rp}: {ms:.4f} ms/step  {n / ms / 1e3:.1f} Mscalar-mul/s", flush=True)
