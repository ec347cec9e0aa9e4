#!/usr/bin/env python3
"""A/B: consecutive bench steps alternate between TWO contexts (own streams, own arenas): the window of one step's MSM (sort, NTT, reduce chain)
then runs beside the other context's accumulation.  python scripts/ab/two_contexts.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
lg = 20; n = 1 << lg
def make(stream):
    b = pkg.Bbg(0)
    b.set_stream(stream.cuda_stream)
    b.set_option("msm_async_reduce", 1)
    srs = b.srs_synth_hashed(0xBB254, n)
    b.ntt_prepare(lg)
    return b, srs
d_sc = torch.from_numpy(pkg.synthetic_scalars(0xBB254 + 3, n).view(np.int64)).cuda()
def run(nctx):
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nctx - 1)]
    ctxs = [make(s) for s in streams]
    d_c = [torch.from_numpy(pkg.synthetic_scalars(0xBB254 + 120, n).view(np.int64)).cuda() for _ in range(nctx)]
    out = [torch.zeros(12, dtype=torch.int64, device="cuda") for _ in range(nctx)]
    torch.cuda.synchronize()
    k = [0]
    def step():
        i = k[0] % nctx; k[0] += 1
        b, srs = ctxs[i]
        b.msm_device(srs, d_sc.data_ptr(), n, out[i].data_ptr())
        b.ntt_device(d_c[i].data_ptr(), lg, 0)
    for _ in range(6): step()
    for b, _ in ctxs: b.join()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20): step()
        for b, _ in ctxs: b.join()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    same = all(np.array_equal(pkg.Bbg.g1_normalize(ctxs[0][0], o.cpu().numpy().view(np.uint64).reshape(1, 12)), pkg.Bbg.g1_normalize(ctxs[0][0], out[0].cpu().numpy().view(np.uint64).reshape(1, 12))) for o in out)
    for b, s in ctxs:
        s.free(); b.close()
    return best, same
for nctx in (1, 2, 3, 1):
    ms, same = run(nctx)
    print(f"{nctx} context(s): {ms:.4f} ms/step  {n / ms / 1e3:.1f} Mscalar-mul/s  same point: {same}", flush=True)
