#!/usr/bin/env python3
"""Where the sharded pipeline's per-step overhead comes from (world of one): the bench step with (a) nothing, (b) bbg.join(1) after each MSM,
(c) join + a 96-byte copy on the main stream, (d) join + event + copy on a side stream."""
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
part = [torch.zeros(12, dtype=torch.int64, device="cuda") for _ in range(2)]
res = [torch.zeros(12, dtype=torch.int64, device="cuda") for _ in range(2)]
bbg.ntt_prepare(lg)
side = torch.cuda.Stream()
ev = [torch.cuda.Event(), torch.cuda.Event()]
def run(mode):
    cnt = [0]
    def step():
        i = cnt[0]; cnt[0] += 1
        bbg.msm_device(srs, d_sc.data_ptr(), n, part[i & 1].data_ptr())
        if mode >= 1 and i >= 1:
            bbg.join(1)
        if mode == 2 and i >= 1:
            res[(i - 1) & 1].copy_(part[(i - 1) & 1])
        if mode == 3 and i >= 1:
            ev[i & 1].record(torch.cuda.current_stream())
            side.wait_event(ev[i & 1])
            with torch.cuda.stream(side):
                res[(i - 1) & 1].copy_(part[(i - 1) & 1])
        bbg.ntt_device(d_c.data_ptr(), lg, 0)
    for _ in range(5): step()
    bbg.join(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20): step()
        bbg.join(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    return best
for mode, name in ((0, "plain step"), (1, "+ join(1)"), (2, "+ join(1) + copy on the main stream"), (3, "+ join(1) + event + copy on a side stream"), (0, "plain step (again)")):
    print(f"{name:45s} {run(mode):.4f} ms/step", flush=True)
