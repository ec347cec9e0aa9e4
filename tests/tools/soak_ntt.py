#!/usr/bin/env python3
"""One-off soak (not collected by pytest): random NTT sizes / ops / generator sizes / constants / pass plans / pass kernels against the oracle,
plus coset_fft_extend and coset_fft_split.  python tests/tools/soak_ntt.py [cases]"""
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
rng = np.random.default_rng(7)
bad = 0
for c in range(cases):
    if c % 8 == 0:  # r5: another pass plan and pass kernel -- the 29-bit-limb kernel (with the constant-operand product in its radix >= 2^9
        # passes: ntt_max_logr8 up to 11 puts such passes into plans of 2^11 .. 2^16 points) against the 32-bit ones
        B.set_option("ntt_max_logr8", int(rng.integers(6, 12)))
        B.set_option("ntt_limbs29", int(rng.integers(-1, 2)))
        B.set_option("ntt_lds_planes", int(rng.integers(0, 3)))
    lg = int(rng.integers(0, 15)) if c % 16 else int(rng.integers(15, 18))
    n = 1 << lg
    op = int(rng.integers(0, 8))
    gs = int(rng.integers(1, n + 1)) if op in (2, 5, 6) and rng.integers(2) else 0
    a = pkg.synthetic_scalars(int(rng.integers(1 << 30)), n)
    if c % 3 == 0:  # un-reduced representatives inside the contract [0, 2r): x + r for every other element
        r_mod = np.array([0x43E1F593F0000001, 0x2833E84879B97091, 0xB85045B68181585D, 0x30644E72E131A029], dtype=np.uint64)
        a = O.canon(0, a)
        for i in range(0, n, 2):
            carry = 0
            for j in range(4):
                t = int(a[i, j]) + int(r_mod[j]) + carry
                a[i, j] = t & 0xFFFFFFFFFFFFFFFF
                carry = t >> 64
    k = pkg.synthetic_scalars(int(rng.integers(1 << 30)), 1)[0] if op >= 4 else None
    got = O.canon(0, B.ntt(a, op, gs, k))
    want = O.canon(0, O.ntt(a, op, gs, k))
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH ntt", lg, op, gs, flush=True)
    if lg >= 1 and c % 4 == 0:
        ld = max(lg + int(rng.integers(0, 3)), 2)
        ext = O.canon(0, B.coset_fft_extend(a, ld))
        z = np.zeros((1 << ld, 4), dtype=np.uint64)
        z[:n] = a
        w = O.canon(0, O.ntt(z, 2, n))
        if ld >= 2 and not (np.array_equal(ext[:-4], w) and np.array_equal(ext[-4:], w[:4])):
            bad += 1
            print("MISMATCH extend", lg, ld, flush=True)
print(f"ntt soak: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
