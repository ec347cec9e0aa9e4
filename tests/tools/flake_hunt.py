#!/usr/bin/env python3
"""One-off: repeat real proofs under several engine / library combinations and count rejected proofs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle.oracle import Oracle, RefProver  # noqa: E402

pkg = ge.load_package()
import torch  # noqa: E402
O = Oracle()
bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
x = O.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
pts = O.srs_powers(x, (2 << 13) + 1)
import callback_engines as E
mode = sys.argv[2] if len(sys.argv) > 2 else "default"
if mode == "ownstream":
    bbg.close()
    bbg = pkg.Bbg(0)  # library-owned (non-default) stream


class SyncRound34(E.Round34Engine):
    def round3_raw(self, *a):
        super().round3_raw(*a)
        torch.cuda.synchronize()

    def round4_raw(self, *a):
        super().round4_raw(*a)
        torch.cuda.synchronize()


variants = [("resident/gpu-lib", E.ResidentEngine, True), ("round346/gpu-lib", E.Round346Engine, True), ("round34/gpu-lib", E.Round34Engine, True),
            ("fused/gpu-lib", E.FusedFftEngine, True), ("shim only/gpu-lib", None, True), ("resident/plain-lib", E.ResidentEngine, False)]
print("mode", mode, flush=True)
for name, cls, linked in variants:
    bad = 0
    for r in range(reps):
        P = RefProver(1 << 13, 12 + r, pts, x, gpu_linked=linked)
        srs = bbg.srs_register(P.monomials())
        P.prove(cls(bbg, srs) if cls is not None else None, check=False)
        if P.verify() != 1:
            bad += 1
        srs.free()
        P.free()
    print(f"{name}: {bad} rejected of {reps}", flush=True)
