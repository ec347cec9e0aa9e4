#!/usr/bin/env python3
"""The reference's REAL provers (compiled from its own sources, oracle/_ref) on an arithmetic circuit, timed four ways
  cpu        ProverBase::construct_proof() as shipped, on the host cores
  link_only  the same call with the fourteen MSM / FFT entry points wrapped (shim/wrap_flags.txt): the round arithmetic stays on the host
  wrapped    the same call in the build that ALSO wraps construct_proof() (shim/bbg_prover_wrap.cpp + wrap_flags_prover.txt): ZERO source
             edits, the driver object is the CPU build's own; first = including the one-off key upload of the circuit, warm = median of 5
  glue       bbg_shim::construct_proof called explicitly (the two-line patch of INTEGRATION.md 2c), for comparison
Every proof is verified with the reference verifier; the wrapped proof on replayed randomness must equal the CPU proof byte for byte.
    python tests/tools/real_prover_r04.py [log2n ...]        (BBG_ALL_FLAVOURS=1: all five prover types)"""
import json
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
sizes = [int(a) for a in sys.argv[1:]] or [16, 18, 20]
x = O.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
pts = O.srs_powers(x, (1 << max(sizes)) + 2)
flavours = ((0, "TurboPLONK"), (1, "StandardPLONK"))
if os.environ.get("BBG_ALL_FLAVOURS"):
    flavours += ((2, "MiMC (Standard + MiMC widget)"), (3, "UnrolledTurbo"), (4, "UnrolledStandard"))
for lg in sizes:
    for flavour, name in flavours:
        gates = (1 << lg) - 64
        A = RefProver(gates, 11, pts, x, flavour=flavour)
        t0 = time.perf_counter(); cpu, blind = A.prove_recording(); t_cpu = time.perf_counter() - t0
        ok_cpu = A.verify(); threads = A.threads
        A.free()
        for timed in (False, True):  # a warm-up session, then a fresh one: the reference body cannot prove twice on one prover object
            W = RefProver(gates, 11, pts, x, wrap_linked=True, flavour=flavour)
            W.wrap_set_enabled(False)
            t0 = time.perf_counter(); W.prove_reference(); t_link = time.perf_counter() - t0
            ok_link = W.verify()
            W.wrap_set_enabled(True)
            W.free()
        W = RefProver(gates, 11, pts, x, wrap_linked=True, flavour=flavour)
        t0 = time.perf_counter(); got = W.prove_reference(replay=blind); t_first = time.perf_counter() - t0
        ok_wrap = W.verify()
        warm = []
        for _ in range(5):  # back to back, like the glue's loop below (a verifier run between two proofs lets the device clock down)
            W.lib.refp_reset(W.h)
            t0 = time.perf_counter(); W.prove_reference(); warm.append(time.perf_counter() - t0)
        assert W.verify() == 1
        W.free(); W.wrap_trim()
        G = RefProver(gates, 11, pts, x, gpu_linked=True, flavour=flavour)
        G.resident_key_create()
        G.prove_resident(blind)
        glue = sorted(G.prove_resident()[1] for _ in range(5))[2]
        G.free()
        print(json.dumps({"prover": name, "log2_gates": lg, "host_threads": threads, "cpu_ms": round(t_cpu * 1e3, 1), "link_only_ms": round(t_link * 1e3, 1),
                          "wrapped_zero_edits_first_ms": round(t_first * 1e3, 1), "wrapped_zero_edits_ms": round(sorted(warm)[2] * 1e3, 2),
                          "explicit_glue_ms": round(glue * 1e3, 2), "byte_identical_to_cpu_proof": got == cpu, "verified": [ok_cpu, ok_link, ok_wrap],
                          "speedup_wrapped_vs_cpu": round(t_cpu / sorted(warm)[2], 1)}), flush=True)
