#!/usr/bin/env python3
"""A/B of the MSM reduce phase: one lane per EC operation (msm_reduce_quad = 0) against four lanes per operation (= 1, default).
Checks both against the oracle, then times stand-alone MSMs (call + sync) and the reduce phase alone (HIP events)."""
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
N = 1 << 20
srs = bbg.srs_synth_hashed(0xBB254, N)
sc = pkg.synthetic_scalars(7, N)
d_sc = torch.from_numpy(sc.view(np.int64).reshape(-1)).cuda()
out = torch.zeros(12, dtype=torch.int64, device="cuda")
pts = srs.read(0, 5000)
for quad in (0, 1):
    bbg.set_option("msm_reduce_quad", 15 if quad else 0)
    for n in (1, 2, 17, 1000, 5000):
        assert np.array_equal(O.jac_to_affine(bbg.msm(srs, sc[:n])), O.pippenger(sc[:n], pts[:n])), (quad, n)
    same = np.tile(sc[:1], (3000, 1))
    assert np.array_equal(O.jac_to_affine(bbg.msm(srs, same)), O.pippenger(same, pts[:3000])), quad
ref = None
print("oracle parity ok for both variants")
print("log2n  quad  standalone_ms  reduce_ms(events)")
for lg in (10, 14, 16, 18, 20):
    n = 1 << lg
    res = []
    for quad in (0, 1):
        bbg.set_option("msm_reduce_quad", 15 if quad else 0)
        for _ in range(3):
            bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
        bbg.sync()
        r = out.cpu().numpy().view(np.uint64)
        res.append(O.jac_to_affine(r))
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
            bbg.sync()
            ts.append(time.perf_counter() - t0)
        bbg.profile_enable(True)
        for _ in range(10):
            bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
        bbg.sync()
        red = bbg.profile_get("msm_reduce")[0] / 10
        bbg.profile_enable(False)
        print(f"{lg:5d}  {quad:4d}  {sorted(ts)[10] * 1e3:13.3f}  {red:10.3f}", flush=True)
    assert np.array_equal(res[0], res[1]), lg
print("both variants give the same points at every size")
