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
pts_dup = pts.copy()
pts_dup[1::2] = pts_dup[0::2]
pts_dup[3::8, 4:] = O.fe_sub(1, np.zeros((len(pts_dup[3::8]), 4), dtype=np.uint64), pts_dup[3::8, 4:])  # some partners negated: P + (-P)
srs_dup = B.srs_register(pts_dup)
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
    dup = c % 7 == 3  # every point twice with the same scalar (or its negative): doublings / cancellations inside bucket runs (k_redo)
    if dup:
        start &= ~1
        n = max(2, n & ~1)
        sc = sc[:n].copy()
        sc[1::2] = sc[0::2]
    B.set_option("msm_window", int(rng.choice([0, 8, 13, 16, 17, 19, 20, 22])))
    B.set_option("msm_accumulate_quad", int(rng.integers(0, 2)))
    B.set_option("msm_limbs29", int(rng.integers(0, 4) != 0))  # mostly the default 29-bit-limb accumulation, sometimes the 32-bit one
    B.set_option("msm_async_reduce", int(rng.integers(0, 2)))
    res = B.msm(srs_dup if dup else srs, sc, start=start)
    want = O.msm_naive(sc, pts_dup[start:start + n]) if dup else O.pippenger(sc, pts[start:start + n])
    if (int(want[3]) >> 63) != 0:
        ok = (int(res[3]) >> 63) != 0
    else:
        ok = np.array_equal(O.jac_to_affine(res), want)
    if not ok:
        bad += 1
        print("MISMATCH case", c, "n", n, "start", start, "kind", kind, flush=True)
print(f"soak: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
