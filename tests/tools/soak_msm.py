#!/usr/bin/env python3
"""One-off soak (not collected by pytest): random MSM sizes / offsets / window widths / sort paths / scalar mixes against the
oracle.  python tests/tools/soak_msm.py [cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

pkg = ge.load_package()
O = Oracle()
B = pkg.Bbg(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(20260928)
N = 1 << 13
srs = B.srs_synth_hashed(0xBB254, N)
pts = srs.read()
bad = 0
for c in range(cases):
    n = int(rng.integers(1, 3000)) if c % 5 else int(rng.integers(3000, N))
    start = int(rng.integers(0, N - n + 1))
    kind = c % 4
    if kind == 0:
        sc = pkg.synthetic_scalars(int(rng.integers(1 << 30)), n)
    elif kind == 1:
        sc = pkg.inputs.mixed_scalars(int(rng.integers(1 << 30)), n, lambda p: O.to_mont(0, p))
    elif kind == 2:  # few distinct scalars: heavy bucket collisions
        base = pkg.synthetic_scalars(int(rng.integers(1 << 30)), 3)
        sc = base[rng.integers(0, 3, n)]
    else:  # raw 256-bit limbs (un-reduced representatives)
        sc = rng.integers(0, 1 << 63, (n, 4), dtype=np.int64).astype(np.uint64) * 2 + 1
    B.set_option("msm_window", int(rng.choice([0, 16, 17, 19, 20, 22])))
    B.set_option("msm_accumulate_quad", int(rng.integers(0, 2)))
    B.set_option("msm_async_reduce", int(rng.integers(0, 2)))
    got = O.jac_to_affine(B.msm(srs, sc, start=start))
    want = O.pippenger(sc, pts[start:start + n])
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH case", c, "n", n, "start", start, "kind", kind, flush=True)
print(f"soak: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
