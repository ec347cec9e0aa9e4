#!/usr/bin/env python3
"""Round 4 A/B of the NTT pass kernels: k_ntt_pass8 (tile resident in LDS between steps, option ntt_lds_planes = 2) against k_ntt_pass8s
(one 16-byte plane at a time through half the LDS, ntt_lds_planes = 1), isolated, HIP events around bursts of in-place transforms, best of
5 bursts; results of the two kernels compared word for word.  BBG_LIB_PATH selects an A/B build of the library (occupancy bound / late fetch
of the output multipliers: csrc/ntt_pass8.hip.h BBG_NTT_OCC, BBG_NTT_LATE_OUTMUL).  Usage: r04_ntt_ab.py [log2n ...]"""
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
sizes = [int(a) for a in sys.argv[1:]] or [16, 18, 20, 21, 22, 24]
tag = os.path.basename(os.environ.get("BBG_LIB_PATH", "libbbg.so"))
for lg in sizes:
    n = 1 << lg
    src = torch.from_numpy(pkg.synthetic_scalars(11, n).view(np.int64).reshape(-1)).cuda()
    outs, times = {}, {}
    for planes in (2, 1, 29):  # 29 = k_ntt_pass29 (option ntt_limbs29)
        bbg.set_option("ntt_lds_planes", planes if planes != 29 else 0)
        bbg.set_option("ntt_limbs29", 1 if planes == 29 else 0)
        row = []
        for op in (0, 2):  # fft, coset_fft
            a = src.clone()
            bbg.ntt_device(a.data_ptr(), lg, op)
            torch.cuda.synchronize()
            outs[(planes, op)] = a.clone()
            best = 1e9
            reps = 50 if lg <= 21 else 12
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    bbg.ntt_device(a.data_ptr(), lg, op)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps)
            row.append(best)
        times[planes] = row
    same = all(torch.equal(outs[(2, op)], outs[(1, op)]) for op in (0, 2))
    # the 29-bit-limb kernel delivers the same residues; its representative (coarse, < 2p) may differ: compare canonical values
    def canon(t):
        a = t.cpu().numpy().view(np.uint64).reshape(-1, 4)
        return bbg.field_op(0, 4, a)
    same29 = all(np.array_equal(canon(outs[(2, op)]), canon(outs[(29, op)])) for op in (0, 2)) if lg <= 22 else None
    print(f"{tag:34s} 2^{lg:2d}  planes2 fft {times[2][0]:.4f} coset {times[2][1]:.4f} | planes1 fft {times[1][0]:.4f} coset {times[1][1]:.4f} | limbs29 fft {times[29][0]:.4f} coset {times[29][1]:.4f} ms | identical {same} {same29}", flush=True)
    assert same and same29 is not False
