#!/usr/bin/env python3
"""Debug aid: where does the resident prover's proof first differ from the reference CPU proof on the same randomness?
Prints, per 32-byte word of the proof, whether the two agree."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
from oracle.oracle import Oracle, RefProver  # noqa: E402

O = Oracle()
flavour = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 9
x = O.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
pts = O.srs_powers(x, (2 << lg) + 2)
A = RefProver(1 << lg, 21 + flavour, pts, x, flavour=flavour)
cpu, blind = A.prove_recording()
B = RefProver(1 << lg, 21 + flavour, pts, x, gpu_linked=True, flavour=flavour)
print("key check", B.resident_check_key())
gpu, secs = B.prove_resident(blind)
print("len", len(cpu), len(gpu), "verify cpu", A.verify(), "gpu", B.verify(), "equal", cpu == gpu)
words = len(cpu) // 32
print("".join("=" if cpu[32 * i:32 * i + 32] == gpu[32 * i:32 * i + 32] else "X" for i in range(words)))
