#!/usr/bin/env python3
"""The wrapped construct_proof() build (oracle/_ref/libbbprover_wrap.so, or the sanitizer build named by BBG_PROVER_WRAP_SO) without torch
in the process (an LD_PRELOADed ASan runtime and torch's loader do not get along): all five prover types at 2^9 gates byte-identical to the
reference CPU prover on replayed blinding scalars, then the key cache walk of tests/test_gpu_parity.py::test_wrapped_construct_proof_key_cache
(two keys alive, LRU eviction under a byte budget, re-upload after eviction, release when the key's last outside owner is gone)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, RefProver  # noqa: E402

O = Oracle()
x = O.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
pts = O.srs_powers(x, (2 << 11) + 2)
for flavour in range(5):
    A = RefProver(1 << 9, 21 + flavour, pts, x, flavour=flavour)
    proof_cpu, blind = A.prove_recording()
    assert A.verify() == 1
    B = RefProver(1 << 9, 21 + flavour, pts, x, wrap_linked=True, flavour=flavour)
    before = B.wrap_stats()
    proof = B.prove_reference(replay=blind)
    assert B.wrap_stats()[0] == before[0] + 1 and B.verify() == 1 and proof == proof_cpu, flavour
    again = B.prove_reference(reset=True)
    assert B.verify() == 1 and again != proof
    A.free(); B.free()
    print("flavour", flavour, "byte-identical through the wrapped construct_proof()", flush=True)

def splitmix(seed, n):
    out = np.empty((n, 4), dtype=np.uint64)
    s = seed
    for i in range(n):
        for k in range(4):
            s = (s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
            z = s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
            out[i, k] = z ^ (z >> 31)
        out[i, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    return out

blind = splitmix(4711, 15)
P1 = RefProver(1 << 11, 91, pts, x, wrap_linked=True, flavour=0)
P2 = RefProver(1 << 10, 92, pts, x, wrap_linked=True, flavour=1)
P1.wrap_clear()
a1 = P1.prove_reference(replay=blind)
one_key = P1.wrap_bytes()
b1 = P2.prove_reference(replay=blind[:12])
assert P1.wrap_cached_keys() == 2
assert P1.prove_reference(replay=blind, reset=True) == a1 and P2.prove_reference(replay=blind[:12], reset=True) == b1
P1.wrap_set_budget(one_key)
assert P1.wrap_cached_keys() == 1
assert P1.prove_reference(replay=blind, reset=True) == a1 and P1.wrap_cached_keys() == 1
P1.wrap_set_budget(1)
assert P2.prove_reference(replay=blind[:12], reset=True) == b1 and P2.wrap_cached_keys() == 1
P1.wrap_set_budget(0)
assert P1.prove_reference(replay=blind, reset=True) == a1 and P1.wrap_cached_keys() == 2
P2.free()
assert P1.wrap_trim() == 1
P1.free()
Q = RefProver(1 << 9, 93, pts, x, wrap_linked=True, flavour=0)
assert Q.wrap_trim() == 0
Q.free()
# round 5: the seven wrapped execute_*_round symbols (a proof driven round by round), a device error inside a resident round (replay on the
# reference rounds) and inside a wrapped construct_proof() (reference body), and a proving key rewritten after its first proof (re-upload)
for flavour in (0, 1, 3):
    A = RefProver(1 << 9, 51 + flavour, pts, x, flavour=flavour)
    proof_cpu, blind_r = A.prove_recording()
    B = RefProver(1 << 9, 51 + flavour, pts, x, wrap_linked=True, flavour=flavour)
    proof, _, q = B.prove_round_by_round(replay=blind_r)
    assert proof == proof_cpu and q == [0] * 7 and B.verify() == 1 and B.wrap_in_progress() == 0, flavour
    assert B.prove_reference(replay=blind_r, reset=True) == proof_cpu
    A.free(); B.free()
    print("flavour", flavour, "byte-identical through the seven wrapped rounds", flush=True)
for fail_round in (1, 4, 6):
    Q = RefProver(1 << 9, 61, pts, x, wrap_linked=True, flavour=0)
    st0 = Q.wrap_stats()
    Q.wrap_fail_round(fail_round)
    proof, _, q = Q.prove_round_by_round()
    assert Q.verify() == 1 and Q.wrap_stats()[1] == st0[1] + 1 and Q.wrap_in_progress() == 0 and q[fail_round] > 0 or fail_round == 5, fail_round
    Q.free()
    Q = RefProver(1 << 9, 61, pts, x, wrap_linked=True, flavour=0)
    Q.prove_reference()
    Q.wrap_fail_round(fail_round)
    Q.prove_reference(reset=True)
    assert Q.verify() == 1
    Q.wrap_fail_round(0)
    Q.free()
print("device errors inside rounds 1 / 4 / 6: proofs completed on the reference rounds / body", flush=True)
A = RefProver(1 << 9, 62, pts, x, flavour=0)
B = RefProver(1 << 9, 62, pts, x, wrap_linked=True, flavour=0)
B.prove_reference()
re0 = B.wrap_reuploads()
A.key_selector_scale3("q_m"); B.key_selector_scale3("q_m")
proof_cpu, blind_r = A.prove_recording()
assert B.prove_reference(replay=blind_r, reset=True) == proof_cpu and B.wrap_reuploads() == re0 + 1
A.free(); B.free()
print("rewritten proving key: re-uploaded, byte-identical", flush=True)
# r6: ONE coefficient at a row the sampled fingerprint does not look at -- found by the full-content check that runs on host threads beside the proof
A = RefProver(1 << 9, 63, pts, x, flavour=0)
B = RefProver(1 << 9, 63, pts, x, wrap_linked=True, flavour=0)
B.prove_reference()
re0 = B.wrap_reuploads()
A.key_selector_poke("q_m", 1); B.key_selector_poke("q_m", 1)
proof_cpu, blind_r = A.prove_recording()
assert B.prove_reference(replay=blind_r, reset=True) == proof_cpu and B.wrap_reuploads() == re0 + 1
assert B.prove_reference(replay=blind_r, reset=True) == proof_cpu and B.wrap_reuploads() == re0 + 1
A.free(); B.free()
print("one poked coefficient of a cached key: found by the full check, proof repeated over the key as it is", flush=True)
print("san_wrap_check PASS")
