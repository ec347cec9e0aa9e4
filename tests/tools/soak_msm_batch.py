#!/usr/bin/env python3
"""One-off soak (not collected by pytest): random BATCHES of MSMs over one SRS through one launch set (bbg_msm_batch) -- 1..8 members, ragged
lengths incl. 0 and 1, per-member offsets, window widths, scalar mixes incl. heavy bucket collisions and P / -P pairs (k_redo) -- every member
against the oracle.  python tests/tools/soak_msm_batch.py [cases]"""
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
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(20260929)
N = 1 << 12
srs = B.srs_synth_hashed(0xBB254, N)
pts = srs.read()
pts_dup = pts.copy()
pts_dup[1::2] = pts_dup[0::2]
pts_dup[3::8, 4:] = O.fe_sub(1, np.zeros((len(pts_dup[3::8]), 4), dtype=np.uint64), pts_dup[3::8, 4:])  # some partners negated: P + (-P)
srs_dup = B.srs_register(pts_dup)
bad = 0
members = 0
for c in range(cases):
    k = int(rng.integers(1, 9))
    dup = c % 5 == 2
    scs, starts = [], []
    for m in range(k):
        r = int(rng.integers(0, 10))
        n = 0 if r == 0 else 1 if r == 1 else int(rng.integers(2, 600)) if r < 7 else int(rng.integers(600, N))
        start = int(rng.integers(0, N - n + 1))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            sc = pkg.synthetic_scalars(int(rng.integers(1 << 30)), max(n, 1))[:n]
        elif kind == 1:  # few distinct scalars: heavy bucket collisions
            base = pkg.synthetic_scalars(int(rng.integers(1 << 30)), 3)
            sc = base[rng.integers(0, 3, n)]
        else:  # raw 256-bit limbs (un-reduced representatives)
            sc = rng.integers(0, 1 << 63, (n, 4), dtype=np.int64).astype(np.uint64) * 2 + 1
        if dup and n >= 2:
            start &= ~1
            n &= ~1
            sc = sc[:n].copy()
            sc[1::2] = sc[0::2]
        scs.append(np.ascontiguousarray(sc.reshape(-1, 4)))
        starts.append(start)
    B.set_option("msm_window", int(rng.choice([0, 8, 13, 16, 17, 19])))
    B.set_option("msm_limbs29", int(rng.integers(0, 4) != 0))
    B.set_option("msm_async_reduce", int(rng.integers(0, 2)))
    res = B.msm_batch(srs_dup if dup else srs, scs, starts=starts)
    for m in range(k):
        n, start, sc = scs[m].shape[0], starts[m], scs[m]
        members += 1
        if n == 0:
            ok = (int(res[m][3]) >> 63) != 0
        else:
            want = O.msm_naive(sc, pts_dup[start:start + n]) if dup else O.pippenger(sc, pts[start:start + n])
            ok = (int(res[m][3]) >> 63) != 0 if (int(want[3]) >> 63) != 0 else np.array_equal(O.jac_to_affine(res[m]), want)
        if not ok:
            bad += 1
            print("MISMATCH case", c, "member", m, "of", k, "n", n, "start", start, flush=True)
print(f"batch soak: {cases} batches, {members} members, {bad} mismatches")
sys.exit(1 if bad else 0)
