#!/usr/bin/env python3
"""Workload for a rocprofv3 --kernel-trace of the small end: 30 batches of four 2^12-term MSMs (bbg_msm_batch_device), each followed by a
sync, then 30 groups of four single MSMs.  Also prints the H2D rate of a 32 MiB pageable host array (what round 1 of a 2^20-gate proof
uploads four times)."""
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
bbg.set_option("msm_async_reduce", 1)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = 1 << lg
srs = bbg.srs_synth_hashed(0xBB254, n)
scs = [torch.from_numpy(pkg.synthetic_scalars(100 + k, n).view(np.int64).reshape(-1)).cuda() for k in range(4)]
ptrs = [s.data_ptr() for s in scs]
res = torch.zeros(4 * 12, dtype=torch.int64, device="cuda")
for rep in range(30):
    bbg.msm_batch_device(srs, ptrs, [n] * 4, res.data_ptr())
    bbg.join(); bbg.sync()
for rep in range(30):
    for k in range(4):
        bbg.msm_device(srs, ptrs[k], n, res.data_ptr() + 96 * k)
    bbg.join(); bbg.sync()
big = pkg.synthetic_scalars(5, 1 << 20)
d = torch.empty(4 << 20, dtype=torch.int64, device="cuda")
import ctypes
for _ in range(2):
    bbg._ck(bbg.lib.bbg_dev_upload(bbg.ctx, ctypes.c_void_p(d.data_ptr()), big.ctypes.data, big.nbytes))
t0 = time.perf_counter()
for _ in range(5):
    bbg._ck(bbg.lib.bbg_dev_upload(bbg.ctx, ctypes.c_void_p(d.data_ptr()), big.ctypes.data, big.nbytes))
dt = (time.perf_counter() - t0) / 5
print(f"H2D 32 MiB pageable: {dt*1e3:.2f} ms = {big.nbytes/dt/1e9:.1f} GB/s", file=sys.stderr)
pin = torch.from_numpy(big.view(np.int64)).pin_memory()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    d.copy_(pin.reshape(-1), non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"H2D 32 MiB pinned:   {dt*1e3:.2f} ms = {big.nbytes/dt/1e9:.1f} GB/s", file=sys.stderr)
