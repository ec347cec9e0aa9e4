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
    # mode 0: one context, one stream (bench.py).  mode 1: the NTT in a context of its own on a second stream, equal priorities.
    # mode 2: NTT stream high priority, MSM main stream low.  (bbg_set_stream synchronises, so the stream is fixed per context.)
    hi, lo = -1, 0
    s_main = torch.cuda.Stream(priority=lo if mode == 2 else 0) if mode != 0 else torch.cuda.current_stream()
    bbg = pkg.Bbg(0)
    bbg.set_stream(s_main.cuda_stream)
    bbg.set_option("msm_reduce_priority", reduce_prio)
    bbg.set_option("msm_async_reduce", 1)
    if mode == 0:
        bn = bbg
    else:
        s_ntt = torch.cuda.Stream(priority=hi if mode == 2 else 0)
        bn = pkg.Bbg(0)
        bn.set_stream(s_ntt.cuda_stream)
    srs = bbg.srs_synth_hashed(0xBB254, n)
    d_sc = torch.from_numpy(pkg.synthetic_scalars(0xBB254 + 3, n).view(np.int64)).cuda()
    d_c = torch.from_numpy(pkg.synthetic_scalars(0xBB254 + 120, n).view(np.int64)).cuda()
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    bn.ntt_prepare(lg)
    torch.cuda.synchronize()
    def step():
        bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
        bn.ntt_device(d_c.data_ptr(), lg, 0)
    for _ in range(5): step()
    bbg.join(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20): step()
        bbg.join(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    ref = out.cpu().numpy().copy()
    if bn is not bbg: bn.close()
    bbg.close()
    return best, ref
res = {}
for mode, name in ((0, "one context, one stream (bench.py)"), (1, "NTT context on its own stream"), (2, "NTT stream high, MSM main stream low"), (0, "one context, one stream (again)")):
    for rp in (1, 0):
        ms, ref = run(mode, rp)
        print(f"{name:40s} msm_reduce_priority={rp}: {ms:.4f} ms/step  {n / ms / 1e3:.1f} Mscalar-mul/s", flush=True)
