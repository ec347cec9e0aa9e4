#!/usr/bin/env python3
"""bbg_msm with host-resident scalars (what the link-time shim calls): one monolithic upload in front of the MSM (msm_upload_pieces = 1)
against the upload in four pieces with each piece's counting pass started as soon as it has landed (= 4).  Measured: pieces lose."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
from oracle.oracle import Oracle  # noqa: E402

O = Oracle()
bbg = pkg.Bbg(0)
N = 1 << 22
srs = bbg.srs_synth_hashed(0xBB254, N)
sc = pkg.synthetic_scalars(7, N)
print("log2n  pieces  host_msm_ms")
for lg in (16, 18, 20, 22):
    n = 1 << lg
    ref = None
    for pieces in (1, 4):
        bbg.set_option("msm_upload_pieces", pieces)
        for _ in range(3):
            out = bbg.msm(srs, sc[:n])
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            out = bbg.msm(srs, sc[:n])
            ts.append(time.perf_counter() - t0)
        aff = O.jac_to_affine(out)
        if ref is None:
            ref = aff
        assert np.array_equal(ref, aff)
        print(f"{lg:5d}  {pieces:6d}  {sorted(ts)[len(ts) // 2] * 1e3:11.3f}", flush=True)
