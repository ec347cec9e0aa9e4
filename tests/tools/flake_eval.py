#!/usr/bin/env python3
"""One-off: shim-wrapped evaluate vs the reference body while a second library context is busy on the GPU."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle.oracle import PROVER_GPU_SO  # noqa: E402

pkg = ge.load_package()
import torch  # noqa: E402
bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
L = ctypes.CDLL(PROVER_GPU_SO, mode=os.RTLD_NOW)
L.refp_diag_evaluate.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
L.refp_diag_evaluate.restype = ctypes.c_int
n = 1 << 14
z = pkg.synthetic_scalars(5, 1)[0]
polys = [torch.from_numpy(pkg.synthetic_scalars(100 + k, 4 * n).view(np.int64).reshape(-1)).cuda() for k in range(21)]
quot = torch.empty(4 * n * 4, dtype=torch.int64, device="cuda")
ch = pkg.synthetic_scalars(7, 9)
for busy in (False, True):
    bad = 0
    for it in range(300):
        c = pkg.synthetic_scalars(1000 + it, n if it % 2 else 4 * n)
        if busy:  # the other context: widget kernels + a synchronous download right before the shim call
            for w in range(5):
                bbg.quotient_widget_device(w, [p.data_ptr() for p in polys], 16, ch, quot.data_ptr())
            quot.cpu()
        r = L.refp_diag_evaluate(c.ctypes.data, c.shape[0], z.ctypes.data)
        assert r >= 0
        bad += (r == 0)
    print(f"other context busy={busy}: {bad} wrong evaluations of 300", flush=True)
