"""Developer probe (not a test, not the bench): shakes the HIP path against the oracle on a GPU box and prints
timings.  Usage: python tests/tools/gpu_probe.py [stage ...]  with stages: field ntt msm time"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
from oracle.oracle import Oracle  # noqa: E402

stages = sys.argv[1:] or ["field", "ntt", "msm", "time"]
O = Oracle()
B = pkg.Bbg(0)
import torch  # noqa: E402
B.set_stream(torch.cuda.current_stream().cuda_stream)  # torch fills and library kernels share one stream
inp = pkg.inputs


def check(name, a, b):
    ok = np.array_equal(a, b)
    print(f"{'ok  ' if ok else 'FAIL'} {name}", flush=True)
    if not ok:
        bad = np.argwhere(np.any(np.asarray(a).reshape(-1, a.shape[-1]) != np.asarray(b).reshape(-1, b.shape[-1]), axis=1))
        print("   first mismatches at rows", bad[:8].ravel(), "of", len(a))
    return ok


if "field" in stages:
    a = inp.synthetic_scalars(1, 4096)
    b = inp.synthetic_scalars(2, 4096)
    a[0] = 0xFFFFFFFFFFFFFFFF
    b[1] = 0
    for which in (0, 1):
        check(f"field{which} mul", B.field_op(which, 0, a, b), O.fe_mul(which, a, b))
        check(f"field{which} mul cios", B.field_op(which, 3, a, b), O.fe_mul(which, a, b))
        check(f"field{which} add", B.field_op(which, 1, a, b), O.fe_add(which, a, b))
        check(f"field{which} sub", B.field_op(which, 2, a, b), O.fe_sub(which, a, b))
        check(f"field{which} from_mont", B.field_op(which, 4, a), O.from_mont(which, a))
        check(f"field{which} to_mont", B.field_op(which, 5, a), O.to_mont(which, a))

if "coarse" in stages:
    P = [np.array([0x43E1F593F0000001, 0x2833E84879B97091, 0xB85045B68181585D, 0x30644E72E131A029], dtype=np.uint64),
         np.array([0x3C208C16D87CFD47, 0x97816a916871ca8d, 0xb85045b68181585d, 0x30644e72e131a029], dtype=np.uint64)]
    def addp(x, p):
        out = x.copy(); carry = np.zeros(len(x), dtype=object)
        for j in range(4):
            t = x[:, j].astype(object) + int(p[j]) + carry
            out[:, j] = np.array([int(v) & 0xFFFFFFFFFFFFFFFF for v in t], dtype=np.uint64)
            carry = np.array([int(v) >> 64 for v in t], dtype=object)
        return out
    for which in (0, 1):
        a = O.canon(which, inp.synthetic_scalars(31, 2000)); b = O.canon(which, inp.synthetic_scalars(32, 2000))
        ap, bp = addp(a, P[which]), addp(b, P[which])
        want = O.fe_mul(which, a, b)
        for name, x, y in (("a,b", a, b), ("a+p,b", ap, b), ("a,b+p", a, bp), ("a+p,b+p", ap, bp)):
            check(f"field{which} raw mul {name}", B.field_op(which, 6, x, y), want)
            check(f"field{which} raw cios {name}", B.field_op(which, 7, x, y), want)
if "ntt" in stages:
    k = inp.synthetic_scalars(77, 1)[0]
    for lg in (0, 1, 2, 3, 5, 8, 11, 12, 13, 14, 16):
        c = inp.synthetic_scalars(100 + lg, 1 << lg)
        for op in range(8):
            kk = k if op >= 4 else None
            got = O.canon(0, B.ntt(c, op, 0, kk))
            check(f"ntt lg={lg} op={op}", got, O.ntt(c, op, 0, kk))
        if lg >= 2:
            gs = (1 << lg) // 4
            for op in (2, 5, 6):
                kk = k if op >= 4 else None
                check(f"ntt lg={lg} op={op} gs={gs}", O.canon(0, B.ntt(c, op, gs, kk)), O.ntt(c, op, gs, kk))
    c = inp.synthetic_scalars(5, 256)
    for ext in (2, 4, 8):
        check(f"coset split ext={ext}", O.canon(0, B.coset_fft_split(c, ext)), O.coset_fft_split(c, ext))

if "msm" in stages:
    n = 1 << 12
    pts_l = O.srs_linear(0x123456789ABCDEF, 0xFEDCBA987654321, n)
    srs_l = B.srs_synth_linear(0x123456789ABCDEF, 0xFEDCBA987654321, n)
    check("srs synth linear", srs_l.read(), pts_l)
    pts_h = O.srs_hashed(0xBB254, n)
    srs_h = B.srs_synth_hashed(0xBB254, n)
    check("srs synth hashed", srs_h.read(), pts_h)
    srs_r = B.srs_register(pts_h)
    check("srs register", srs_r.read(), pts_h)
    sc = inp.synthetic_scalars(3, n)
    for nn in (0, 1, 2, 17, 100, 1000, n):
        got = O.jac_to_affine(B.msm(srs_h, sc[:nn]))
        check(f"msm n={nn}", got, O.pippenger(sc[:nn], pts_h[:nn]))
    got = O.jac_to_affine(B.msm(srs_l, sc[:1000], start=100))
    check("msm linear from=100", got, O.pippenger(sc[:1000], pts_l[100:1100]))
    mixed = inp.mixed_scalars(9, n, lambda p: O.to_mont(0, p))
    check("msm mixed", O.jac_to_affine(B.msm(srs_h, mixed)), O.pippenger(mixed, pts_h))
    same = np.tile(sc[:1], (n, 1))
    check("msm all-equal scalars", O.jac_to_affine(B.msm(srs_h, same)), O.pippenger(same, pts_h))

if "time" in stages:
    import torch
    for lg in (16, 18, 20, 22, 24):
        n = 1 << lg
        c = inp.synthetic_scalars(lg, n)
        t = torch.from_numpy(c.view(np.int64)).cuda()
        B.ntt_prepare(lg)
        B.ntt_device(t.data_ptr(), lg, 0)
        B.sync()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            B.ntt_device(t.data_ptr(), lg, 0)
        B.sync()
        dt = (time.perf_counter() - t0) / reps
        print(f"ntt 2^{lg}: {dt*1e3:.3f} ms  {1.5*n*lg/dt/1e9:.2f} Gfield-op/s  {64*n/dt/1e9:.1f} GB/s(alg)", flush=True)
    for lg in (16, 20):
        n = 1 << lg
        t0 = time.perf_counter()
        srs = B.srs_synth_hashed(0xBB254, n)
        print(f"srs synth+tables 2^{lg}: {time.perf_counter()-t0:.3f} s")
        sc = inp.synthetic_scalars(1234, n)
        ts = torch.from_numpy(sc.view(np.int64)).cuda()
        out = torch.zeros(12, dtype=torch.int64, device="cuda")
        B.msm_device(srs, ts.data_ptr(), n, out.data_ptr())
        B.sync()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            B.msm_device(srs, ts.data_ptr(), n, out.data_ptr())
        B.sync()
        dt = (time.perf_counter() - t0) / reps
        print(f"msm 2^{lg}: {dt*1e3:.3f} ms  {n/dt/1e6:.2f} Mscalar-mul/s", flush=True)
        if lg == 16:
            pts = srs.read()
            got = O.jac_to_affine(out.cpu().numpy().view(np.uint64))
            check("msm 2^16 vs oracle", got, O.pippenger(sc, pts))
        srs.free()
if "msmsweep" in stages:
    import torch
    for lg in (10, 12, 14, 16, 18, 20, 22, 24):
        n = 1 << lg
        srs = B.srs_synth_hashed(0xBB254, n)
        sc = inp.synthetic_scalars(1234, n)
        ts = torch.from_numpy(sc.view(np.int64)).cuda()
        out = torch.zeros(12, dtype=torch.int64, device="cuda")
        for mode in (0, 1):
            B.set_option("msm_async_reduce", mode)
            B.msm_device(srs, ts.data_ptr(), n, out.data_ptr()); B.join(); B.sync()
            reps = 10 if lg <= 20 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                B.msm_device(srs, ts.data_ptr(), n, out.data_ptr())
            B.join(); B.sync()
            dt = (time.perf_counter() - t0) / reps
            B.profile_enable(True)
            for _ in range(3):
                B.msm_device(srs, ts.data_ptr(), n, out.data_ptr())
            B.join(); B.sync()
            ph = {k[4:]: round(B.profile_get(k)[0] / 3, 4) for k in ("msm_recode", "msm_sort", "msm_accumulate", "msm_reduce")}
            B.profile_enable(False)
            print(f"msm 2^{lg} async_reduce={mode}: {dt*1e3:.3f} ms  {n/dt/1e6:.2f} Mscalar-mul/s  phases {ph}", flush=True)
        B.set_option("msm_async_reduce", 0)
        srs.free()
if "msmwin" in stages:
    import torch
    for lg in (19, 20, 21, 22):
        n = 1 << lg
        srs = B.srs_synth_hashed(0xBB254, n)
        ts = torch.from_numpy(inp.synthetic_scalars(1234, n).view(np.int64)).cuda()
        out = torch.zeros(12, dtype=torch.int64, device="cuda")
        B.set_option("msm_async_reduce", 1)
        for win in (16, 20):
            B.set_option("msm_window", win)
            B.msm_device(srs, ts.data_ptr(), n, out.data_ptr()); B.join(); B.sync()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                B.msm_device(srs, ts.data_ptr(), n, out.data_ptr())
            B.join(); B.sync()
            dt = (time.perf_counter() - t0) / reps
            print(f"msm 2^{lg} window={win} pipelined: {dt*1e3:.3f} ms  {n/dt/1e6:.2f} Mscalar-mul/s", flush=True)
        B.set_option("msm_window", 0)
        B.set_option("msm_async_reduce", 0)
        srs.free()
if "msmab" in stages:
    import torch
    for lg in (20, 21):
        n = 1 << lg
        srs = B.srs_synth_hashed(0xBB254, n)
        ts = torch.from_numpy(inp.synthetic_scalars(1234, n).view(np.int64)).cuda()
        out = torch.zeros(12, dtype=torch.int64, device="cuda")
        B.set_option("msm_async_reduce", 1)
        res = {16: [], 20: []}
        for rnd in range(6):
            for win in (16, 20):
                B.set_option("msm_window", win)
                for _ in range(3):
                    B.msm_device(srs, ts.data_ptr(), n, out.data_ptr())
                B.join(); B.sync()
                reps = 20
                t0 = time.perf_counter()
                for _ in range(reps):
                    B.msm_device(srs, ts.data_ptr(), n, out.data_ptr())
                B.join(); B.sync()
                res[win].append((time.perf_counter() - t0) / reps * 1e3)
        for win in (16, 20):
            print(f"msm 2^{lg} window={win} pipelined, 6 interleaved rounds x 20: " + " ".join(f"{x:.3f}" for x in res[win]) + f"  min {min(res[win]):.3f} ms", flush=True)
        B.set_option("msm_window", 0)
        B.set_option("msm_async_reduce", 0)
        srs.free()
if "hostpath" in stages:
    import torch, ctypes
    for lg in (18, 20, 22):
        n = 1 << lg
        srs = B.srs_synth_hashed(0xBB254, n)
        sc = inp.synthetic_scalars(1234, n)
        pinned = torch.from_numpy(sc.view(np.int64).copy()).pin_memory()
        out = np.zeros(12, dtype=np.uint64)
        def run(ptr):
            B._ck(B.lib.bbg_msm(B.ctx, srs.handle, ctypes.c_void_p(ptr), 0, n, out.ctypes.data))
        for name, ptr in (("pageable", sc.ctypes.data), ("pinned", pinned.data_ptr())):
            run(ptr)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); run(ptr); ts.append(time.perf_counter() - t0)
            print(f"bbg_msm host path 2^{lg} ({name} scalars): best {min(ts)*1e3:.3f} ms  median {sorted(ts)[2]*1e3:.3f} ms", flush=True)
        c = inp.synthetic_scalars(5, n)
        pc = torch.from_numpy(c.view(np.int64).copy()).pin_memory()
        for name, ptr in (("pageable", c.ctypes.data), ("pinned", pc.data_ptr())):
            B._ck(B.lib.bbg_ntt(B.ctx, ctypes.c_void_p(ptr), lg, 0, 0, None))
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); B._ck(B.lib.bbg_ntt(B.ctx, ctypes.c_void_p(ptr), lg, 0, 0, None)); ts.append(time.perf_counter() - t0)
            print(f"bbg_ntt host path 2^{lg} ({name} coeffs, in place): best {min(ts)*1e3:.3f} ms  median {sorted(ts)[2]*1e3:.3f} ms", flush=True)
        srs.free()
if "quotient" in stages:
    import torch
    names = ["permutation", "turbo_arithmetic", "turbo_fixed_base", "turbo_range", "turbo_logic"]
    reads = [12, 13, 18, 7, 10]  # 32-byte values touched per point (distinct arrays + shifted rows + quotient read/write)
    for lg in (20, 22, 24):
        m = 1 << lg
        polys = [torch.from_numpy(inp.synthetic_scalars(100 + k, m).view(np.int64).reshape(-1)).cuda() for k in range(21)]
        quot = torch.zeros(m * 4, dtype=torch.int64, device="cuda")
        ptrs = [p.data_ptr() for p in polys]
        ch = inp.synthetic_scalars(7, 9)
        B.ntt_prepare(lg)
        tot = 0.0
        for w in range(5):
            B.quotient_widget_device(w, ptrs, lg, ch, quot.data_ptr()); B.sync()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                B.quotient_widget_device(w, ptrs, lg, ch, quot.data_ptr())
            B.sync()
            dt = (time.perf_counter() - t0) / reps
            tot += dt
            print(f"quotient widget {names[w]:18s} 4n=2^{lg}: {dt*1e3:.3f} ms  {reads[w]*32*m/dt/1e9:.0f} GB/s ({reads[w]} x 32 B per point)", flush=True)
        print(f"all five widgets 4n=2^{lg}: {tot*1e3:.3f} ms", flush=True)
        del polys, quot
if "grandproduct" in stages:
    import torch
    for lg in (16, 20, 22):
        n = 1 << lg
        dw = [torch.from_numpy(inp.synthetic_scalars(800 + k, n).view(np.int64).reshape(-1)).cuda() for k in range(4)]
        ds = [torch.from_numpy(inp.synthetic_scalars(810 + k, n).view(np.int64).reshape(-1)).cuda() for k in range(4)]
        z = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
        ch = inp.synthetic_scalars(820, 5)
        B.ntt_prepare(lg)
        f = lambda: B.permutation_grand_product_device([t.data_ptr() for t in dw], [t.data_ptr() for t in ds], lg, ch[0], ch[1], ch[2:5], z.data_ptr())
        f(); B.sync()
        t0 = time.perf_counter()
        for _ in range(5): f()
        B.sync()
        dt = (time.perf_counter() - t0) / 5
        print(f"permutation grand product n=2^{lg}: {dt*1e3:.3f} ms  ({n/dt/1e6:.0f} M rows/s)", flush=True)
if "tune" in stages:
    import torch
    for lg in (20, 22, 24):
        n = 1 << lg
        c = inp.synthetic_scalars(lg, n)
        t = torch.from_numpy(c.view(np.int64)).cuda()
        for tile, maxr in ((12, 9), (12, 10), (12, 8), (12, 7), (11, 9), (11, 7), (10, 7)):
            B.set_option("ntt_tile_log", tile)
            B.set_option("ntt_max_logr", maxr)
            B.ntt_prepare(lg)
            B.ntt_device(t.data_ptr(), lg, 0)
            B.sync()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                B.ntt_device(t.data_ptr(), lg, 0)
            B.sync()
            dt = (time.perf_counter() - t0) / reps
            print(f"ntt 2^{lg} tile={tile} maxr={maxr}: {dt*1e3:.3f} ms  {1.5*n*lg/dt/1e9:.2f} Gfield-op/s", flush=True)
if "nttprof" in stages:
    import torch
    for lg in (20, 24):
        n = 1 << lg
        t = torch.from_numpy(inp.synthetic_scalars(lg, n).view(np.int64)).cuda()
        B.ntt_prepare(lg)
        for _ in range(5):
            B.ntt_device(t.data_ptr(), lg, 0)
        B.sync()
if "polytime" in stages:
    import torch
    for lg in (22, 24):
        n = 1 << lg
        a = torch.from_numpy(inp.synthetic_scalars(1, n).view(np.int64).reshape(-1)).cuda()
        b = torch.from_numpy(inp.synthetic_scalars(2, n).view(np.int64).reshape(-1)).cuda()
        r = torch.empty_like(a)
        z = inp.synthetic_scalars(3, 1)[0]
        def timeit(fn, reps=10):
            fn(); B.sync()
            t0 = time.perf_counter()
            for _ in range(reps): fn()
            B.sync()
            return (time.perf_counter() - t0) / reps
        for op, name in ((0, "add"), (1, "sub"), (2, "mul")):
            dt = timeit(lambda: B.poly_op_device(op, a.data_ptr(), b.data_ptr(), r.data_ptr(), n))
            print(f"poly {name} 2^{lg}: {dt*1e3:.3f} ms  {96*n/dt/1e9:.0f} GB/s algorithmic (96 B/elem) = {96*n/dt/8e12*100:.1f}% of 8 TB/s", flush=True)
        dt = timeit(lambda: B.poly_evaluate_device(a.data_ptr(), n, z), 5)
        print(f"evaluate 2^{lg}: {dt*1e3:.3f} ms  {32*n/dt/1e9:.0f} GB/s (32 B/elem)", flush=True)
        dt = timeit(lambda: B.kate_opening_device(a.data_ptr(), r.data_ptr(), n, z), 5)
        print(f"kate opening 2^{lg}: {dt*1e3:.3f} ms  {64*n/dt/1e9:.0f} GB/s (64 B/elem)", flush=True)
        dt = timeit(lambda: B.divide_by_pseudo_vanishing_device(a.data_ptr(), lg - 2, lg, 4))
        print(f"divide_by_pseudo_vanishing 2^{lg}: {dt*1e3:.3f} ms  {64*n/dt/1e9:.0f} GB/s (64 B/elem)", flush=True)
if "tune8" in stages:
    import torch
    for lg in (18, 20, 22, 24):
        n = 1 << lg
        c = inp.synthetic_scalars(lg, n)
        t = torch.from_numpy(c.view(np.int64)).cuda()
        for maxr8 in (7, 8, 9, 10, 11):
            B.set_option("ntt_max_logr8", maxr8)
            B.ntt_prepare(lg)
            B.ntt_device(t.data_ptr(), lg, 0)
            B.sync()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                B.ntt_device(t.data_ptr(), lg, 0)
            B.sync()
            dt = (time.perf_counter() - t0) / reps
            print(f"ntt8 2^{lg} maxr8={maxr8}: {dt*1e3:.3f} ms  {1.5*n*lg/dt/1e9:.2f} Gfield-op/s", flush=True)
print("probe done")
