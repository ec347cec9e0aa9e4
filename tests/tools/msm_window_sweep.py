#!/usr/bin/env python3
"""Window-width sweep behind msm_auto_window (csrc/msm.hip): for every n, every compiled width -- stand-alone MSM (call + sync),
pipelined MSM (a burst of calls, reduce phase of call i beside call i+1) and, at 2^20, the bench step (MSM + NTT).  Rounds are
interleaved over the widths (each width is timed `ROUNDS` times in turn, the median is reported) so that clock drift hits all alike.
All widths must produce the same point (checked through the product's own normalisation).

    python tests/tools/msm_window_sweep.py [lo_log2n hi_log2n [widths...]]
"""
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
LO = int(sys.argv[1]) if len(sys.argv) > 1 else 18
HI = int(sys.argv[2]) if len(sys.argv) > 2 else 24
WIDTHS = [int(a) for a in sys.argv[3:]] or [16, 17, 19, 20, 22]
ROUNDS = 5
out = torch.zeros(12, dtype=torch.int64, device="cuda")
print("log2n  window  standalone_ms  pipelined_ms  step_ms(MSM+NTT)  Mscalar/s(pipelined)  same_point")
for lg in range(LO, HI + 1):
    n = 1 << lg
    srs = bbg.srs_synth_hashed(0xBB254, n)  # the SRS a prover of this size holds: n points
    d_sc = torch.from_numpy(pkg.synthetic_scalars(7, n).view(np.int64).reshape(-1)).cuda()
    d_c = torch.from_numpy(pkg.synthetic_scalars(8, n).view(np.int64).reshape(-1)).cuda()
    bbg.ntt_prepare(lg)
    burst = 16 if lg <= 20 else (6 if lg <= 22 else 3)
    widths = [w for w in WIDTHS if not (w == 16 and lg >= 23)]  # 16 x n x 64 B of tables: skip where it only wastes the budget
    sa, pl, stp, pts = {w: [] for w in widths}, {w: [] for w in widths}, {w: [] for w in widths}, {}
    for w in widths:  # build each width's tables, warm up
        bbg.set_option("msm_window", w)
        bbg.set_option("msm_async_reduce", 0)
        bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
        bbg.sync()
        pts[w] = bbg.g1_normalize(out.cpu().numpy().view(np.uint64).reshape(1, 12))
    for _ in range(ROUNDS):
        for w in widths:
            bbg.set_option("msm_window", w)
            bbg.set_option("msm_async_reduce", 0)
            bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr()); bbg.sync()
            t0 = time.perf_counter()
            bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr()); bbg.sync()
            sa[w].append(time.perf_counter() - t0)
            bbg.set_option("msm_async_reduce", 1)
            for _ in range(2):
                bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
            bbg.join(); bbg.sync()
            t0 = time.perf_counter()
            for _ in range(burst):
                bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
            bbg.join(); bbg.sync()
            pl[w].append((time.perf_counter() - t0) / burst)
            if lg <= 21:
                t0 = time.perf_counter()
                for _ in range(burst):
                    bbg.msm_device(srs, d_sc.data_ptr(), n, out.data_ptr())
                    bbg.ntt_device(d_c.data_ptr(), lg, 0)
                bbg.join(); bbg.sync()
                stp[w].append((time.perf_counter() - t0) / burst)
    med = lambda v: sorted(v)[len(v) // 2] * 1e3 if v else float("nan")
    for w in widths:
        same = "yes" if np.array_equal(pts[w], pts[widths[0]]) else "DIFFERENT"
        print(f"{lg:5d}  {w:6d}  {med(sa[w]):13.3f}  {med(pl[w]):12.3f}  {med(stp[w]):16.3f}  {n / med(pl[w]) / 1e3:20.1f}  {same}", flush=True)
    bbg.set_option("msm_window", 0)
    bbg.set_option("msm_async_reduce", 0)
    srs.free()
    del d_sc, d_c
