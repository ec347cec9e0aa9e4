#!/usr/bin/env python3
"""Isolated NTT-family timings (device resident, in place, HIP events over 50 launches) per size -- the table in profiles/*_sweeps.txt."""
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
sizes = [int(a) for a in sys.argv[1:]] or [12, 14, 16, 18, 20, 22, 24]
print("log2n      fft_ms     ifft_ms  coset_fft_ms  coset_ifft_ms  fft_Gfop/s")
for lg in sizes:
    n = 1 << lg
    a = torch.from_numpy(pkg.synthetic_scalars(11, n).view(np.int64).reshape(-1)).cuda()
    row = []
    for op in (0, 1, 2, 3):  # binding.FFT, IFFT, COSET_FFT, COSET_IFFT
        for _ in range(3):
            bbg.ntt_device(a.data_ptr(), lg, op)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                bbg.ntt_device(a.data_ptr(), lg, op)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50)
        row.append(best)
    print(f"{lg:5d}  {row[0]:10.4f}  {row[1]:10.4f}  {row[2]:12.4f}  {row[3]:13.4f}  {1.5 * n * lg / row[0] / 1e6:10.1f}", flush=True)
