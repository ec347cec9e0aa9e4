#!/usr/bin/env python3
"""Round-2 measurement sweep (supplementary evidence; bench.py is the contract line): device-resident timings of every entry point
family on one MI355X, medians of 20 after warm-up, plus the PCIe-inclusive host-buffer path.  Writes plain text to stdout."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

pkg = ge.load_package()
bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
MASK = torch.tensor([-1, -1, -1, 0x0FFFFFFFFFFFFFFF], dtype=torch.int64, device=dev)


def rand_fr(n, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return (torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=dev, generator=g) & MASK).reshape(-1)


def med(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    bbg.join(); bbg.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        bbg.join(); bbg.sync()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


print("# MSM, device-resident scalars, hashed SRS of 2^24 points; standalone = call + sync; pipelined = 16 calls back to back (async reduce)")
print("log2n  standalone_ms  pipelined_ms  Mscalar/s(pipelined)")
srs = bbg.srs_synth_hashed(0xBB254, 1 << 24)
sc = rand_fr(1 << 24, 1)
out = torch.zeros(12, dtype=torch.int64, device=dev)
for lg in (10, 12, 14, 16, 18, 20, 21, 22, 23, 24):
    n = 1 << lg
    bbg.set_option("msm_async_reduce", 0)
    sa = med(lambda: bbg.msm_device(srs, sc.data_ptr(), n, out.data_ptr()), reps=10)
    bbg.set_option("msm_async_reduce", 1)

    def burst():
        for _ in range(16 if lg <= 20 else 4):
            bbg.msm_device(srs, sc.data_ptr(), n, out.data_ptr())
    pl = med(burst, reps=5) / (16 if lg <= 20 else 4)
    print(f"{lg:5d}  {sa:13.3f}  {pl:12.3f}  {n / pl / 1e3:10.1f}", flush=True)
bbg.set_option("msm_async_reduce", 0)
srs.free()
del sc

print("\n# NTT family, device resident, in place (isolated); Gfield-op/s = 1.5 n log2 n / t")
print("log2n      fft_ms     ifft_ms  coset_fft_ms  coset_ifft_ms  fft_Gfop/s  fft_HBM_frac(64n/t/8TB/s)")
for lg in (12, 14, 16, 18, 20, 22, 24):
    n = 1 << lg
    a = rand_fr(n, 2)
    bbg.ntt_prepare(lg)
    t = [med(lambda: bbg.ntt_device(a.data_ptr(), lg, op)) for op in (0, 1, 2, 3)]
    print(f"{lg:5d}  {t[0]:10.4f}  {t[1]:10.4f}  {t[2]:12.4f}  {t[3]:13.4f}  {1.5 * n * lg / t[0] / 1e6:10.1f}  {64.0 * n / (t[0] * 1e-3) / 8e12:10.4f}", flush=True)
    del a

print("\n# the prover's FFT work item (n coefficients -> 4n coset values, zero-extended read, fused g^j) and polynomial helpers, n = 2^20")
n, lg = 1 << 20, 20
lib = bbg.lib
a = rand_fr(n, 3)
big = torch.zeros((4 * n + 4) * 4, dtype=torch.int64, device=dev)
z = pkg.synthetic_scalars(9, 1)[0]
h = ctypes.c_void_p()
srs = bbg.srs_synth_hashed(0xBB254, n)
gens = pkg.synthetic_scalars(3, 4)
bbg._ck(lib.bbg_prover_create(bbg.ctx, srs.handle, lg, 4, gens.ctypes.data, ctypes.byref(h)))
t_eval = med(lambda: bbg.poly_evaluate_device(a.data_ptr(), n, z))
dst = torch.zeros_like(a)
t_kate = med(lambda: bbg.kate_opening_device(a.data_ptr(), dst.data_ptr(), n, z))
q = rand_fr(4 * n, 4)
t_dpv = med(lambda: bbg.divide_by_pseudo_vanishing_device(q.data_ptr(), lg, lg + 2, 4))
polys = [rand_fr(n, 10 + k) for k in range(14)]
scal = pkg.synthetic_scalars(5, 14)
t_lin = med(lambda: bbg.poly_linear_combination_device([p_.data_ptr() for p_ in polys], scal, a.data_ptr(), dst.data_ptr(), n))
t_mul = med(lambda: bbg.poly_op_device(2, q.data_ptr(), q.data_ptr(), q.data_ptr(), 4 * n))
print(f"evaluate(n)            {t_eval:8.4f} ms  {32.0 * n / t_eval / 1e9:7.2f} TB/s   (1 product per 32 B: multiplier-bound at 4.5 TB/s; includes the host sync)")
print(f"kate_opening(n)        {t_kate:8.4f} ms  {4 * 32.0 * n / t_kate / 1e9:7.2f} TB/s   (3 reads + 1 write; 3 products per coefficient: multiplier-bound at ~1.5 TB/s)")
print(f"divide_by_Z*_H(4n)     {t_dpv:8.4f} ms  {2 * 32.0 * 4 * n / t_dpv / 1e9:7.2f} TB/s   (read + write; 6 products per value)")
print(f"lincomb 14 terms (n)   {t_lin:8.4f} ms  {16 * 32.0 * n / t_lin / 1e9:7.2f} TB/s   (15 reads + 1 write; 14 products per 512 B)")
print(f"pointwise sqr in place (4n) {t_mul:8.4f} ms  {2 * 32.0 * 4 * n / t_mul / 1e9:7.2f} TB/s   (1 read + 1 write)")
lib.bbg_prover_destroy(h)

print("\n# PCIe-inclusive host-buffer path (what the link-time shim calls), n = 2^20")
hs = pkg.synthetic_scalars(1, n)
t = time.perf_counter(); r = bbg.msm(srs, hs); t_first = (time.perf_counter() - t) * 1e3
ts = []
for _ in range(10):
    t = time.perf_counter(); bbg.msm(srs, hs); ts.append((time.perf_counter() - t) * 1e3)
print(f"bbg_msm(2^20 host scalars)          {sorted(ts)[5]:8.3f} ms")
hc = pkg.synthetic_scalars(2, n)
vp = ctypes.c_void_p
ts = []
for _ in range(10):  # straight through the C ABI on preallocated host arrays (no binding copies)
    t = time.perf_counter(); bbg._ck(lib.bbg_ntt(bbg.ctx, vp(hc.ctypes.data), lg, 1, 0, None)); ts.append((time.perf_counter() - t) * 1e3)
print(f"bbg_ntt(ifft 2^20, in place)        {sorted(ts)[5]:8.3f} ms")
outbuf = np.zeros((4 * n + 4, 4), dtype=np.uint64)
ts = []
for _ in range(6):
    t = time.perf_counter(); bbg._ck(lib.bbg_coset_fft_extend(bbg.ctx, vp(hc.ctypes.data), lg, lg + 2, vp(outbuf.ctypes.data))); ts.append((time.perf_counter() - t) * 1e3)
print(f"bbg_coset_fft_extend(2^20 -> 2^22)  {sorted(ts)[3]:8.3f} ms   (32 MiB up, 128 MiB down)")
srs.free()
