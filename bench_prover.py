#!/usr/bin/env python3
"""Prover-shaped workload (BASELINE.json config 4): the MSM + FFT work of ONE TurboPLONK proof of a
2^20-gate circuit, in the order work_queue::process_queue issues it (prover.cpp:420-436):

    preamble : 4 x iNTT(n)            wires w_1..w_4                       (prover.cpp:152-194)
    round 1  : 4 x MSM(n)             W_1..W_4                             (prover.cpp:66-73)
    round 3  : 1 x iNTT(n) (z), 1 x MSM(n) (Z), 5 x coset-NTT(4n, generator_size n)   (permutation widget, prover.cpp:258-267)
    round 4  : 1 x coset-iNTT(4n) (quotient), 4 x MSM(n)  T_1..T_4         (prover.cpp:304-357)
    round 6  : 2 x MSM(n)             PI_Z, PI_Z_OMEGA                     (kate_commitment_scheme.cpp:134-236)

= 11 MSM(n) + 5 coset-NTT(4n) + 1 coset-iNTT(4n) + 5 iNTT(n), all on polynomials resident in HBM (no PCIe traffic
between items), n = 2^log2n.  The widget / quotient arithmetic between the items is out of scope and is
not emulated: this measures the hot path only, like the "7.5 s" anchor in BASELINE.md section 2.

    python bench_prover.py [--log2n 20] [--reps 5] [--cpu]     # --cpu also times the reference binary on the host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
IFFT, COSET_FFT, COSET_IFFT = 1, 2, 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    lg = args.log2n
    n = 1 << lg
    dev = torch.device("cuda", 0)
    bbg = pkg.Bbg(0)
    bbg.set_stream(torch.cuda.current_stream().cuda_stream)
    bbg.set_option("msm_async_reduce", 1)
    srs = bbg.srs_synth_hashed(0xBB254, n)
    polys = [torch.from_numpy(pkg.synthetic_scalars(0xBB254 + 500 + i, n).view(np.int64)).to(dev) for i in range(11)]
    big = [torch.zeros(4 * n * 4, dtype=torch.int64, device=dev) for _ in range(6)]
    for i in range(6):
        big[i][: 4 * n] = polys[i % 11].reshape(-1)  # n non-zero coefficients on the 4n domain (proving_key.cpp:21-22)
    outs = torch.zeros(11 * 12, dtype=torch.int64, device=dev)
    bbg.ntt_prepare(lg)
    bbg.ntt_prepare(lg + 2)

    def msm(i):
        bbg.msm_device(srs, polys[i].data_ptr(), n, outs[12 * i:].data_ptr())

    def proof():
        for i in range(4):
            bbg.ntt_device(polys[i].data_ptr(), lg, IFFT)
        for i in range(4):
            msm(i)
        bbg.ntt_device(polys[4].data_ptr(), lg, IFFT)
        msm(4)
        for i in range(5):
            bbg.ntt_device(big[i].data_ptr(), lg + 2, COSET_FFT, n)
        bbg.ntt_device(big[5].data_ptr(), lg + 2, COSET_IFFT)
        for i in range(5, 9):
            msm(i)
        for i in range(9, 11):
            msm(i)

    # ---- the same sequence with the O(n) round arithmetic between the transforms on the device as well:
    # round 3 grand product, round 4 five quotient widgets + pseudo-vanishing division, round 5 evaluations (14 openings + t),
    # round 6 two opening-polynomial accumulations (~20 + ~5 terms) and two Kate quotients
    sel = [torch.from_numpy(pkg.synthetic_scalars(0xBB254 + 600 + i, 4 * n).view(np.int64).reshape(-1)).to(dev) for i in range(16)]
    ch9 = pkg.synthetic_scalars(0xBB254 + 650, 9)
    zeta = pkg.synthetic_scalars(0xBB254 + 651, 1)[0]
    nu = pkg.synthetic_scalars(0xBB254 + 652, 25)
    z_lag = torch.zeros(n * 4, dtype=torch.int64, device=dev)
    open1 = torch.zeros(n * 4, dtype=torch.int64, device=dev)
    open2 = torch.zeros(n * 4, dtype=torch.int64, device=dev)
    kate_out = torch.zeros(n * 4, dtype=torch.int64, device=dev)

    def proof_full():
        for i in range(4):
            bbg.ntt_device(polys[i].data_ptr(), lg, IFFT)
        for i in range(4):
            msm(i)
        bbg.permutation_grand_product_device([polys[i].data_ptr() for i in range(4)], [polys[5 + i].data_ptr() for i in range(4)], lg,
                                             ch9[2], ch9[3], ch9[6:9], z_lag.data_ptr())
        bbg.ntt_device(z_lag.data_ptr(), lg, IFFT)
        msm(4)
        for i in range(5):
            bbg.ntt_device(big[i].data_ptr(), lg + 2, COSET_FFT, n)
        ptrs = [big[i].data_ptr() for i in range(5)] + [t.data_ptr() for t in sel]
        ab = ch9[0].copy()
        for w in range(5):
            c = ch9.copy()
            c[0] = ab
            ab = bbg.quotient_widget_device(w, ptrs, lg + 2, c, big[5].data_ptr())
        bbg.divide_by_pseudo_vanishing_device(big[5].data_ptr(), lg, lg + 2, 4)
        bbg.ntt_device(big[5].data_ptr(), lg + 2, COSET_IFFT)
        for i in range(5, 9):
            msm(i)
        for i in range(11):  # opening evaluations at zeta (and zeta * omega for the shifted ones), t(zeta) over 4n coefficients
            bbg.poly_evaluate_device(polys[i].data_ptr(), n, zeta)
        for i in range(3):
            bbg.poly_evaluate_device(polys[i].data_ptr(), n, zeta)
        bbg.poly_evaluate_device(big[5].data_ptr(), 4 * n, zeta)
        bbg.poly_linear_combination_device([polys[i % 11].data_ptr() for i in range(20)], nu[:20], big[5].data_ptr(), open1.data_ptr(), n)
        bbg.poly_linear_combination_device([polys[i].data_ptr() for i in range(5)], nu[20:25], None, open2.data_ptr(), n)
        bbg.kate_opening_device(open1.data_ptr(), kate_out.data_ptr(), n, zeta)
        bbg.kate_opening_device(open2.data_ptr(), open1.data_ptr(), n, zeta)
        for i in range(9, 11):
            msm(i)

    proof_full()
    torch.cuda.synchronize()
    full = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        proof_full()
        torch.cuda.synchronize()
        full.append(time.perf_counter() - t0)

    proof()
    torch.cuda.synchronize()
    times = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        proof()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best, med = min(times), sorted(times)[len(times) // 2]
    out = {"workload": "TurboPLONK-shaped MSM+FFT sequence, n=2^%d: 11 MSM(n) + 5 coset-NTT(4n) + 1 coset-iNTT(4n) + 5 iNTT(n), HBM resident" % lg,
           "gpu_ms_best": round(best * 1e3, 3), "gpu_ms_median": round(med * 1e3, 3), "n_gpus": 1,
           "with_round_arithmetic": {"what": "+ grand product, 5 quotient widgets, pseudo-vanishing division, 15 evaluations, 2 opening "
                                             "accumulations (20 + 5 terms), 2 Kate quotients, all HBM resident",
                                     "gpu_ms_best": round(min(full) * 1e3, 3), "gpu_ms_median": round(sorted(full)[len(full) // 2] * 1e3, 3)},
           "reference_anchor_ms": 7500.0 if lg == 20 else None,
           "anchor_note": "BASELINE.md section 2: reference binary on an 8-vCPU Xeon (survey container), different machine"}
    if args.cpu:
        from oracle.oracle import Ref, ref_available
        if ref_available():
            ref = Ref()
            pts = srs.read()
            ctx = ref.msm(pts)
            sc = pkg.synthetic_scalars(0xBB254 + 500, n)
            co = pkg.synthetic_scalars(0xBB254 + 501, 4 * n)
            _, t_msm = ctx.run(sc)
            _, t_msm = ctx.run(sc)
            ctx.free()
            d1, d4 = ref.domain(lg, 0), ref.domain(lg + 2, n)
            _, t_ifft = d1.run(sc, 1)
            _, t_cfft = d4.run(co, 2)
            _, t_cifft = d4.run(co, 3)
            d1.free(); d4.free()
            cpu = 11 * t_msm + 5 * t_cfft + t_cifft + 5 * t_ifft
            out["cpu_reference_ms"] = round(cpu * 1e3, 1)
            out["cpu_cores"] = ref.num_threads()
            out["cpu_detail_ms"] = {"msm": round(t_msm * 1e3, 1), "ifft_n": round(t_ifft * 1e3, 2), "coset_fft_4n": round(t_cfft * 1e3, 1),
                                    "coset_ifft_4n": round(t_cifft * 1e3, 1)}
            out["gpu_over_cpu_same_box"] = round(cpu / best, 1)
    print(json.dumps(out))
    srs.free()
    bbg.close()


if __name__ == "__main__":
    main()
