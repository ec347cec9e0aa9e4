#!/usr/bin/env python3
"""Soak: many resident proofs (fresh randomness, varying circuits, both flavours, sessions created and destroyed, several proofs per
key) -- every one must be accepted by the reference verifier.  Catches stream-ordering and lifetime bugs that a single proof hides."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
from oracle.oracle import Oracle, RefProver  # noqa: E402

O = Oracle()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
x = O.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
pts = O.srs_powers(x, (2 << 14) + 2)
bad = 0
t0 = time.time()
for i in range(rounds):
    flavour = i % 5  # Turbo, Standard, MiMC, UnrolledTurbo, UnrolledStandard
    lg = 9 + (i % 6)
    # MiMCComposer (flavour 2) only at (2^lg - 64) gates: at sizes just below a power of two the reference's own MiMC composer corrupts
    # its heap while building the witness (its CPU prover aborts on the same circuit: "free(): invalid next size") -- not a device matter
    gates = (1 << lg) - 64 if flavour == 2 else (1 << lg) - (i % 7)
    P = RefProver(gates, 1000 + i, pts, x, gpu_linked=True, flavour=flavour)
    for rep in range(1 + (i % 3)):
        proof, secs = P.prove_resident()
        ok = P.verify()
        if ok != 1:
            bad += 1
            print("REJECTED", i, flavour, lg, rep, flush=True)
    P.free()
print(f"soak: {rounds} sessions, rejected proofs: {bad}, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
