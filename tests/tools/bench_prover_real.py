#!/usr/bin/env python3
"""End-to-end check of the drop-in seam with the REAL reference prover (test tooling: it drives the checker under oracle/, so it
lives under tests/; not a bench.py line, supplementary evidence).

Runs the reference's TurboPLONK prover (oracle/ref_prover_driver.cpp, prebuilt into oracle/_ref/libbbprover.so from the
reference's own sources) over the same circuit twice:

  cpu : work_queue::process_queue as shipped (pippenger_unsafe / coset_fft / ifft on the host cores)
  gpu : the same work items handed to this library's C-ABI host entry points (bbg_msm / bbg_ntt: host buffers in and out,
        PCIe-inclusive; the SRS is registered once, like the Pippenger constructor does)

and reports wall-clock split into "rounds" (the prover's own widget / transcript / polynomial logic, always on the CPU) and
"queue" (the MSM + FFT work items).  Both proofs are verified with the reference's TurboVerifier.

    python tests/tools/bench_prover_real.py [--log2n 16] [--check]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import callback_engines  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=16, help="circuit size after padding")
    ap.add_argument("--check", action="store_true", help="compare every GPU work item with the reference CPU result")
    args = ap.parse_args()
    pkg = ge.load_package()
    from oracle.oracle import Oracle, RefProver, prover_available
    if not prover_available():
        raise SystemExit("oracle/_ref/libbbprover.so not available (make -C oracle prover, needs /root/reference)")
    O = Oracle()
    n = 1 << args.log2n
    x = O.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
    t0 = time.perf_counter()
    pts = O.srs_powers(x, n + 1)
    t_srs = time.perf_counter() - t0
    gates = n - 64  # the composer pads to the next power of two

    P = RefProver(gates, 11, pts, x)
    assert P.n == n, (P.n, n)
    t0 = time.perf_counter()
    proof_cpu = P.prove()
    t_cpu = time.perf_counter() - t0
    ok_cpu = P.verify()
    cpu = {"total_ms": round(t_cpu * 1e3, 1), "rounds_ms": round(P.t_rounds * 1e3, 1), "queue_ms": round(P.t_queue * 1e3, 1)}
    P.free()

    bbg = pkg.Bbg(0)
    P = RefProver(gates, 11, pts, x)
    srs = bbg.srs_register(P.monomials())

    import ctypes

    eng = callback_engines.FusedFftEngine(bbg, srs)
    warm = np.zeros((4 * n, 4), dtype=np.uint64)  # warm-up: scratch allocation, twiddle tables, window tables
    wout = np.zeros(12, dtype=np.uint64)
    eng.msm_raw(warm.ctypes.data, n, wout.ctypes.data)
    eng.coset_fft_raw(warm.ctypes.data, args.log2n + 2, n)
    eng.ifft_raw(warm.ctypes.data, args.log2n)
    t0 = time.perf_counter()
    proof_gpu = P.prove(eng, check=args.check)
    t_gpu = time.perf_counter() - t0
    ok_gpu = P.verify()
    gpu = {"total_ms": round(t_gpu * 1e3, 1), "rounds_ms": round(P.t_rounds * 1e3, 1), "queue_ms": round(P.t_queue * 1e3, 1),
           "round_ms": [round(t * 1e3, 1) for t in P.t_round],
           "items": {"msm": P.counts[0], "coset_fft_4n": P.counts[1], "ifft_n": P.counts[2]},
           "mismatching_items": P.mismatches if args.check else None}
    # + round 4's quotient (five widgets, divide_by_pseudo_vanishing, coset_ifft) on the device; selectors resident per key
    P4 = RefProver(gates, 11, pts, x)
    eng4 = callback_engines.Round346Engine(bbg, srs)
    P4.prove(eng4, check=False)  # warm-up proof: uploads the per-key arrays
    P4.free()
    P4 = RefProver(gates, 11, pts, x)
    # NOTE: a new RefProver = a new proving key at new addresses, so the per-key arrays are uploaded again inside this proof
    t0 = time.perf_counter()
    P4.prove(eng4, check=False)
    t4 = time.perf_counter() - t0
    gpu4 = {"total_ms": round(t4 * 1e3, 1), "rounds_ms": round(P4.t_rounds * 1e3, 1), "queue_ms": round(P4.t_queue * 1e3, 1),
            "round_ms": [round(t * 1e3, 1) for t in P4.t_round], "verified": P4.verify() == 1}
    P4.free()
    linked = both = resident = None
    from oracle.oracle import PROVER_GPU_SO
    if os.path.exists(PROVER_GPU_SO):  # the unmodified prover with the shim linked in front (no callbacks, inline FFTs included)
        PL = RefProver(gates, 11, pts, x, gpu_linked=True)
        PL.prove()  # warm-up proof: context, twiddles, scratch
        PL.free()
        PL = RefProver(gates, 11, pts, x, gpu_linked=True)
        t0 = time.perf_counter()
        PL.prove()
        t_l = time.perf_counter() - t0
        # shim-linked prover (queue + inline helpers through --wrap) AND round 4 taken over by the engine
        eng5 = callback_engines.Round346Engine(bbg, None)
        eng5.queue_via_reference = True
        PB = RefProver(gates, 11, pts, x, gpu_linked=True)
        PB.prove(eng5, check=False)  # warm-up: uploads the per-key arrays
        PB.free()
        PB = RefProver(gates, 11, pts, x, gpu_linked=True)
        t0 = time.perf_counter()
        PB.prove(eng5, check=False)
        t_b = time.perf_counter() - t0
        both = {"total_ms": round(t_b * 1e3, 1), "rounds_ms": round(PB.t_rounds * 1e3, 1), "queue_ms": round(PB.t_queue * 1e3, 1),
                "round_ms": [round(t * 1e3, 1) for t in PB.t_round], "verified": PB.verify() == 1}
        PB.free()
        # shim-linked prover for the inline helpers, work queue through the callbacks, FFT results resident for round 4
        eng6 = callback_engines.ResidentEngine(bbg, srs)
        PR = RefProver(gates, 11, pts, x, gpu_linked=True)
        PR.prove(eng6, check=False)
        PR.free()
        PR = RefProver(gates, 11, pts, x, gpu_linked=True)
        t0 = time.perf_counter()
        PR.prove(eng6, check=False)
        t_r = time.perf_counter() - t0
        resident = {"total_ms": round(t_r * 1e3, 1), "rounds_ms": round(PR.t_rounds * 1e3, 1), "queue_ms": round(PR.t_queue * 1e3, 1),
                    "round_ms": [round(t * 1e3, 1) for t in PR.t_round], "verified": PR.verify() == 1}
        PR.free()
        linked = {"total_ms": round(t_l * 1e3, 1), "rounds_ms": round(PL.t_rounds * 1e3, 1), "queue_ms": round(PL.t_queue * 1e3, 1),
                  "round_ms": [round(t * 1e3, 1) for t in PL.t_round], "verified": PL.verify() == 1}
        PL.free()
    out = {"workload": f"reference TurboProver, arithmetic circuit, n = 2^{args.log2n} gates after padding",
           "host_threads": P.threads, "srs_setup_s": round(t_srs, 2),
           "cpu_engine": cpu, "gpu_engine": gpu, "gpu_engine_plus_rounds346": gpu4, "gpu_shim_linked": linked, "gpu_shim_linked_plus_rounds346": both, "gpu_shim_linked_rounds346_resident_ffts": resident, "proof_bytes": len(proof_gpu),
           "verified": {"cpu": ok_cpu == 1, "gpu": ok_gpu == 1},
           "queue_speedup": round(cpu["queue_ms"] / max(gpu["queue_ms"], 1e-9), 1),
           "end_to_end_speedup": round(cpu["total_ms"] / max(gpu["total_ms"], 1e-9), 2)}
    print(json.dumps(out))
    assert ok_cpu == 1 and ok_gpu == 1 and len(proof_cpu) == len(proof_gpu)
    srs.free()
    P.free()
    bbg.close()


if __name__ == "__main__":
    main()
