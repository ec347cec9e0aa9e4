#!/usr/bin/env python3
"""The reference's REAL TurboPLONK / StandardPLONK provers (compiled from its own sources, oracle/_ref) on an arithmetic circuit, timed
  cpu       ProverBase::construct_proof() as shipped, on the host cores
  shim      the same call in the build linked with shim/bbg_barretenberg_shim.cpp + --wrap (no source change)
  resident  bbg_shim::construct_proof (shim/bbg_resident_prover.hpp): every O(n) step on the device
Every proof is verified with the reference verifier; the resident proof on replayed randomness must equal the CPU proof byte for byte.
    python tests/tools/real_prover_r02.py [log2n ...]"""
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
for lg in sizes:
    flavours = ((0, "TurboPLONK"), (1, "StandardPLONK"))
    if os.environ.get("BBG_ONLY_FLAVOURS"):
        names = {0: "TurboPLONK", 1: "StandardPLONK", 2: "MiMC", 3: "UnrolledTurbo", 4: "UnrolledStandard", 6: "TurboPLONK arithmetic-only"}
        flavours = tuple((int(f), names[int(f)]) for f in os.environ["BBG_ONLY_FLAVOURS"].split(","))
    elif os.environ.get("BBG_ALL_FLAVOURS"):  # + MiMCComposer's prover and the unrolled provers (what the rollup circuits use)
        flavours += ((2, "MiMC (Standard + MiMC widget)"), (3, "UnrolledTurbo"), (4, "UnrolledStandard"),
                     (6, "TurboPLONK, arithmetic gates only"))
    for flavour, name in flavours:
        gates = (1 << lg) - 64
        A = RefProver(gates, 11, pts, x, flavour=flavour)
        t0 = time.perf_counter(); cpu, blind = A.prove_recording(); t_cpu = time.perf_counter() - t0
        ok_cpu = A.verify(); threads = A.threads
        A.free()
        S = RefProver(gates, 11, pts, x, gpu_linked=True, flavour=flavour)
        S.prove_reference()  # warm-up: SRS window tables, twiddles, scratch
        S.free()
        S = RefProver(gates, 11, pts, x, gpu_linked=True, flavour=flavour)
        t0 = time.perf_counter(); S.prove_reference(); t_shim = time.perf_counter() - t0
        ok_shim = S.verify()
        S.free()
        B = RefProver(gates, 11, pts, x, gpu_linked=True, flavour=flavour)
        t_key = B.resident_key_create()
        gpu, t_first = B.prove_resident(blind)
        ok_gpu = B.verify()
        warm = []
        for _ in range(5):
            _, t = B.prove_resident()
            warm.append(t)
            assert B.verify() == 1
        B.free()
        print(json.dumps({"prover": name, "log2_gates": lg, "host_threads": threads, "cpu_ms": round(t_cpu * 1e3, 1), "shim_linked_ms": round(t_shim * 1e3, 1),
                          "resident_ms": round(sorted(warm)[2] * 1e3, 2), "resident_first_ms": round(t_first * 1e3, 2), "resident_key_once_ms": round(t_key * 1e3, 1),
                          "byte_identical_to_cpu_proof": gpu == cpu, "verified": [ok_cpu, ok_shim, ok_gpu],
                          "speedup_resident_vs_cpu": round(t_cpu / sorted(warm)[2], 1)}), flush=True)
