"""ctypes access to the CHECKERS -- test infrastructure only.

  Oracle : oracle/libbn254_oracle.so, this repo's plain-C restatement (bn254_oracle.c).
  Ref    : oracle/_ref/libbbref.so, the real barretenberg hot path compiled from /root/reference
           (ref_driver.cpp + the reference's own TUs); present only where it was built / shipped prebuilt.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libbn254_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libbbref.so")

vp, sz, cint, cu = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint


def build(with_ref=True):
    subprocess.run(["make", "-C", _HERE, "libbn254_oracle.so"], check=True)
    if with_ref:
        subprocess.run(["make", "-C", _HERE, "ref"], check=True)


def _arr(a, last):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim != 2:
        a = a.reshape(-1, last)
    assert a.shape[-1] == last, (a.shape, last)
    return a


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build(with_ref=False)
        L = self.lib = ctypes.CDLL(ORACLE_SO)
        for name, args in {
            "oracle_fe_mul": [cint, vp, vp, vp, sz], "oracle_fe_add": [cint, vp, vp, vp, sz],
            "oracle_fe_sub": [cint, vp, vp, vp, sz], "oracle_fe_inv": [cint, vp, vp, sz],
            "oracle_fe_to_mont": [cint, vp, vp, sz], "oracle_fe_from_mont": [cint, vp, vp, sz],
            "oracle_fe_canon": [cint, vp, vp, sz], "oracle_fr_root_of_unity": [cu, vp],
            "oracle_endo_split": [vp, vp, sz], "oracle_g1_generator": [vp], "oracle_g1_mul": [vp, vp, vp],
            "oracle_g1_add": [vp, vp, vp], "oracle_g1_jac_to_affine": [vp, vp], "oracle_g1_sum": [vp, sz, vp],
            "oracle_g1_to_buffer": [vp, vp], "oracle_srs_linear": [ctypes.c_uint64, ctypes.c_uint64, sz, vp],
            "oracle_srs_powers": [vp, sz, vp], "oracle_srs_hashed": [ctypes.c_uint64, sz, vp], "oracle_point_table": [vp, sz, vp],
            "oracle_wnaf_schedule": [vp, sz, sz, vp, vp, vp], "oracle_pippenger": [vp, vp, sz, vp],
            "oracle_msm_naive": [vp, vp, sz, vp], "oracle_poly_eval": [vp, sz, vp, vp],
        }.items():
            getattr(L, name).argtypes = args
            getattr(L, name).restype = None
        L.oracle_g1_on_curve.argtypes = [vp]; L.oracle_g1_on_curve.restype = cint
        L.oracle_optimal_bucket_width.argtypes = [sz]; L.oracle_optimal_bucket_width.restype = sz
        L.oracle_ntt.argtypes = [vp, cu, cint, sz, vp]; L.oracle_ntt.restype = cint
        L.oracle_coset_fft_split.argtypes = [vp, cu, sz]; L.oracle_coset_fft_split.restype = cint
        L.oracle_num_threads.argtypes = []; L.oracle_num_threads.restype = cint
        L.oracle_set_threads.argtypes = [cint]; L.oracle_set_threads.restype = None
        # (the oracle caps its own OpenMP teams at 32 with num_threads() clauses; the process-wide setting is never touched)
        L.oracle_poly_binop.argtypes = [cint, vp, vp, vp, sz]; L.oracle_poly_binop.restype = None
        L.oracle_kate_opening.argtypes = [vp, vp, sz, vp, vp]; L.oracle_kate_opening.restype = None
        L.oracle_divide_by_pseudo_vanishing.argtypes = [vp, cu, cu, sz]; L.oracle_divide_by_pseudo_vanishing.restype = cint
        L.oracle_permutation_z.argtypes = [vp, vp, cu, vp, vp, vp, vp]
        L.oracle_quotient_widget.argtypes = [cint, vp, cu, vp, vp, vp]; L.oracle_quotient_widget.restype = cint

    # ---- fields (which: 0 Fr, 1 Fq)
    def _bin(self, fn, which, a, b):
        a, b = _arr(a, 4), _arr(b, 4)
        r = np.empty_like(a)
        fn(which, a.ctypes.data, b.ctypes.data, r.ctypes.data, a.shape[0])
        return r

    def _un(self, fn, which, a):
        a = _arr(a, 4)
        r = np.empty_like(a)
        fn(which, a.ctypes.data, r.ctypes.data, a.shape[0])
        return r

    def fe_mul(self, which, a, b): return self._bin(self.lib.oracle_fe_mul, which, a, b)
    def fe_add(self, which, a, b): return self._bin(self.lib.oracle_fe_add, which, a, b)
    def fe_sub(self, which, a, b): return self._bin(self.lib.oracle_fe_sub, which, a, b)
    def fe_inv(self, which, a): return self._un(self.lib.oracle_fe_inv, which, a)
    def to_mont(self, which, a): return self._un(self.lib.oracle_fe_to_mont, which, a)
    def from_mont(self, which, a): return self._un(self.lib.oracle_fe_from_mont, which, a)
    def canon(self, which, a): return self._un(self.lib.oracle_fe_canon, which, a)

    def root_of_unity(self, log2n):
        r = np.empty(4, dtype=np.uint64)
        self.lib.oracle_fr_root_of_unity(log2n, r.ctypes.data)
        return r

    def endo_split(self, scalars):
        s = _arr(scalars, 4)
        r = np.empty_like(s)
        self.lib.oracle_endo_split(s.ctypes.data, r.ctypes.data, s.shape[0])
        return r

    # ---- group
    def g1_generator(self):
        r = np.empty(8, dtype=np.uint64)
        self.lib.oracle_g1_generator(r.ctypes.data)
        return r

    def g1_on_curve(self, p):
        p = np.ascontiguousarray(p, dtype=np.uint64)
        return bool(self.lib.oracle_g1_on_curve(p.ctypes.data))

    def g1_mul(self, p, k):
        p, k = np.ascontiguousarray(p, dtype=np.uint64), np.ascontiguousarray(k, dtype=np.uint64)
        r = np.empty(8, dtype=np.uint64)
        self.lib.oracle_g1_mul(p.ctypes.data, k.ctypes.data, r.ctypes.data)
        return r

    def g1_add(self, p, q):
        p, q = np.ascontiguousarray(p, dtype=np.uint64), np.ascontiguousarray(q, dtype=np.uint64)
        r = np.empty(8, dtype=np.uint64)
        self.lib.oracle_g1_add(p.ctypes.data, q.ctypes.data, r.ctypes.data)
        return r

    def jac_to_affine(self, jac):
        j = np.ascontiguousarray(jac, dtype=np.uint64)
        r = np.empty(8, dtype=np.uint64)
        self.lib.oracle_g1_jac_to_affine(j.ctypes.data, r.ctypes.data)
        return r

    def g1_sum(self, jacs):
        j = _arr(jacs, 12)
        r = np.empty(8, dtype=np.uint64)
        self.lib.oracle_g1_sum(j.ctypes.data, j.shape[0], r.ctypes.data)
        return r

    def g1_to_buffer(self, p):
        p = np.ascontiguousarray(p, dtype=np.uint64)
        buf = np.empty(64, dtype=np.uint8)
        self.lib.oracle_g1_to_buffer(p.ctypes.data, buf.ctypes.data)
        return bytes(buf)

    def srs_linear(self, a, s, n):
        r = np.empty((n, 8), dtype=np.uint64)
        self.lib.oracle_srs_linear(a, s, n, r.ctypes.data)
        return r

    def srs_hashed(self, seed, n):
        r = np.empty((n, 8), dtype=np.uint64)
        self.lib.oracle_srs_hashed(seed, n, r.ctypes.data)
        return r

    def srs_powers(self, x_mont, n):
        x = np.ascontiguousarray(x_mont, dtype=np.uint64)
        r = np.empty((n, 8), dtype=np.uint64)
        self.lib.oracle_srs_powers(x.ctypes.data, n, r.ctypes.data)
        return r

    def point_table(self, points):
        p = _arr(points, 8)
        r = np.empty((2 * p.shape[0], 8), dtype=np.uint64)
        self.lib.oracle_point_table(p.ctypes.data, p.shape[0], r.ctypes.data)
        return r

    # ---- MSM
    def optimal_bucket_width(self, n): return int(self.lib.oracle_optimal_bucket_width(n))

    def wnaf_schedule(self, scalars, wnaf_bits):
        s = _arr(scalars, 4)
        n = s.shape[0]
        rounds = (127 + wnaf_bits - 1) // wnaf_bits
        sched = np.empty((rounds, 2 * n), dtype=np.uint64)
        skew = np.empty(2 * n, dtype=np.uint8)
        counts = np.zeros(256, dtype=np.uint64)
        self.lib.oracle_wnaf_schedule(s.ctypes.data, n, wnaf_bits, sched.ctypes.data, skew.ctypes.data, counts.ctypes.data)
        return sched, skew, counts[:rounds]

    def pippenger(self, scalars, points):
        s, p = _arr(scalars, 4), _arr(points, 8)
        assert s.shape[0] == p.shape[0]
        r = np.empty(8, dtype=np.uint64)
        self.lib.oracle_pippenger(s.ctypes.data, p.ctypes.data, s.shape[0], r.ctypes.data)
        return r

    def msm_naive(self, scalars, points):
        s, p = _arr(scalars, 4), _arr(points, 8)
        assert s.shape[0] == p.shape[0]
        r = np.empty(8, dtype=np.uint64)
        self.lib.oracle_msm_naive(s.ctypes.data, p.ctypes.data, s.shape[0], r.ctypes.data)
        return r

    # ---- NTT
    def ntt(self, coeffs, op=0, generator_size=0, constant=None):
        a = _arr(coeffs, 4).copy()
        n = a.shape[0]
        log2n = n.bit_length() - 1
        assert (1 << log2n) == n
        c = None if constant is None else np.ascontiguousarray(constant, dtype=np.uint64)
        rc = self.lib.oracle_ntt(a.ctypes.data, log2n, op, generator_size, None if c is None else c.ctypes.data)
        assert rc == 0, rc
        return a

    def coset_fft_split(self, coeffs, ext):
        a = _arr(coeffs, 4)
        n = a.shape[0]
        buf = np.zeros((n * ext, 4), dtype=np.uint64)
        buf[:n] = a
        rc = self.lib.oracle_coset_fft_split(buf.ctypes.data, n.bit_length() - 1, ext)
        assert rc == 0, rc
        return buf

    def poly_eval(self, coeffs, z):
        a, z = _arr(coeffs, 4), np.ascontiguousarray(z, dtype=np.uint64)
        r = np.empty(4, dtype=np.uint64)
        self.lib.oracle_poly_eval(a.ctypes.data, a.shape[0], z.ctypes.data, r.ctypes.data)
        return r

    def num_threads(self): return int(self.lib.oracle_num_threads())

    def poly_binop(self, op, a, b):
        a, b = _arr(a, 4), _arr(b, 4)
        r = np.empty_like(a)
        self.lib.oracle_poly_binop(op, a.ctypes.data, b.ctypes.data, r.ctypes.data, a.shape[0])
        return r

    def kate_opening(self, src, z):
        a, z = _arr(src, 4), np.ascontiguousarray(z, dtype=np.uint64)
        dest, f = np.empty_like(a), np.empty(4, dtype=np.uint64)
        self.lib.oracle_kate_opening(a.ctypes.data, dest.ctypes.data, a.shape[0], z.ctypes.data, f.ctypes.data)
        return dest, f

    def quotient_widget(self, widget, polys, log2_large, challenges, quotient):
        """polys: list of 21 (m, 4) arrays in bbg_quotient_poly order (23 for widget 7 = MiMC: + q_mimc_coefficient, q_mimc_selector);
        challenges (9, 4); quotient (m, 4) updated in place.  Returns the next alpha_base (canonical)."""
        arrs = [np.ascontiguousarray(p, dtype=np.uint64) for p in polys]
        assert len(arrs) >= (23 if widget == 7 else 21)
        ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        ch = np.ascontiguousarray(challenges, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        assert quotient.flags["C_CONTIGUOUS"] and quotient.dtype == np.uint64
        rc = self.lib.oracle_quotient_widget(widget, ptrs, log2_large, ch.ctypes.data, quotient.ctypes.data, out.ctypes.data)
        assert rc == 0, rc
        return out

    def permutation_z(self, wires, sigmas, beta, gamma, ks):
        """wires, sigmas: (4, n, 4) Lagrange-base values; ks: (3, 4) coset generators.  Returns z (n, 4), canonical."""
        w = np.ascontiguousarray(wires, dtype=np.uint64)
        s_ = np.ascontiguousarray(sigmas, dtype=np.uint64)
        n = w.shape[1]
        z = np.empty((n, 4), dtype=np.uint64)
        b, g, k = (np.ascontiguousarray(v, dtype=np.uint64) for v in (beta, gamma, ks))
        self.lib.oracle_permutation_z(w.ctypes.data, s_.ctypes.data, n.bit_length() - 1, b.ctypes.data, g.ctypes.data, k.ctypes.data,
                                      z.ctypes.data)
        return z

    def divide_by_pseudo_vanishing(self, evals, log2_src, cut=4):
        a = _arr(evals, 4).copy()
        rc = self.lib.oracle_divide_by_pseudo_vanishing(a.ctypes.data, log2_src, a.shape[0].bit_length() - 1, cut)
        assert rc == 0
        return a


def ref_available():
    if not os.path.exists(REF_SO):
        return False
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return all(f in flags for f in (" adx", " bmi2", " avx2"))


class Ref:
    """The real reference (barretenberg) hot path; see oracle/ref_driver.cpp."""

    def __init__(self):
        if not ref_available():
            raise RuntimeError("oracle/_ref/libbbref.so not available on this machine")
        L = self.lib = ctypes.CDLL(REF_SO)
        L.ref_num_threads.restype = cint
        L.ref_fe_op.argtypes = [cint, cint, vp, vp, vp, sz]
        L.ref_fr_root_of_unity.argtypes = [cu, vp]
        L.ref_fr_coset_generator.argtypes = [vp]
        L.ref_endo_split.argtypes = [vp, vp, sz]
        L.ref_g1_generator.argtypes = [vp]
        L.ref_g1_on_curve.argtypes = [vp]; L.ref_g1_on_curve.restype = cint
        L.ref_g1_mul.argtypes = [vp, vp, vp]
        L.ref_g1_add.argtypes = [vp, vp, vp]
        L.ref_g1_dbl.argtypes = [vp, vp]
        L.ref_g1_to_buffer.argtypes = [vp, vp]
        L.ref_point_table.argtypes = [vp, sz, vp]
        L.ref_wnaf_schedule.argtypes = [vp, sz, vp, vp, vp]; L.ref_wnaf_schedule.restype = cint
        L.ref_msm_new.argtypes = [vp, sz]; L.ref_msm_new.restype = vp
        L.ref_msm_free.argtypes = [vp]
        L.ref_msm_run.argtypes = [vp, vp, sz, sz, cint, vp]; L.ref_msm_run.restype = ctypes.c_double
        L.ref_msm_run_jac.argtypes = [vp, vp, sz, sz, vp]
        L.ref_msm_naive.argtypes = [vp, vp, sz, vp]
        L.ref_g1_sum.argtypes = [vp, sz, vp]
        L.ref_domain_new.argtypes = [cu, sz]; L.ref_domain_new.restype = vp
        L.ref_domain_free.argtypes = [vp]
        L.ref_ntt_run.argtypes = [vp, vp, cint, vp]; L.ref_ntt_run.restype = ctypes.c_double
        L.ref_coset_fft_split.argtypes = [vp, cu, sz]
        L.ref_poly_eval.argtypes = [vp, sz, vp, vp]
        L.ref_poly_binop.argtypes = [cint, vp, vp, vp, cu]
        L.ref_kate_opening.argtypes = [vp, vp, sz, vp, vp]
        L.ref_divide_by_pseudo_vanishing.argtypes = [vp, cu, cu, sz]

    def num_threads(self): return int(self.lib.ref_num_threads())

    def set_threads(self, n):
        self.lib.ref_set_threads.argtypes = [cint]
        self.lib.ref_set_threads(int(n))

    def fe_op(self, which, op, a, b=None):
        a = _arr(a, 4)
        r = np.empty_like(a)
        bp = None
        if b is not None:
            b = _arr(b, 4)
            bp = b.ctypes.data
        self.lib.ref_fe_op(which, op, a.ctypes.data, bp, r.ctypes.data, a.shape[0])
        return r

    def root_of_unity(self, log2n):
        r = np.empty(4, dtype=np.uint64)
        self.lib.ref_fr_root_of_unity(log2n, r.ctypes.data)
        return r

    def coset_generator(self):
        r = np.empty(4, dtype=np.uint64)
        self.lib.ref_fr_coset_generator(r.ctypes.data)
        return r

    def endo_split(self, scalars):
        s = _arr(scalars, 4)
        r = np.empty_like(s)
        self.lib.ref_endo_split(s.ctypes.data, r.ctypes.data, s.shape[0])
        return r

    def g1_generator(self):
        r = np.empty(8, dtype=np.uint64)
        self.lib.ref_g1_generator(r.ctypes.data)
        return r

    def g1_on_curve(self, p):
        p = np.ascontiguousarray(p, dtype=np.uint64)
        return bool(self.lib.ref_g1_on_curve(p.ctypes.data))

    def g1_mul(self, p, k):
        p, k = np.ascontiguousarray(p, dtype=np.uint64), np.ascontiguousarray(k, dtype=np.uint64)
        r = np.empty(8, dtype=np.uint64)
        self.lib.ref_g1_mul(p.ctypes.data, k.ctypes.data, r.ctypes.data)
        return r

    def g1_add(self, p, q):
        p, q = np.ascontiguousarray(p, dtype=np.uint64), np.ascontiguousarray(q, dtype=np.uint64)
        r = np.empty(8, dtype=np.uint64)
        self.lib.ref_g1_add(p.ctypes.data, q.ctypes.data, r.ctypes.data)
        return r

    def g1_dbl(self, p):
        p = np.ascontiguousarray(p, dtype=np.uint64)
        r = np.empty(8, dtype=np.uint64)
        self.lib.ref_g1_dbl(p.ctypes.data, r.ctypes.data)
        return r

    def g1_to_buffer(self, p):
        p = np.ascontiguousarray(p, dtype=np.uint64)
        buf = np.empty(64, dtype=np.uint8)
        self.lib.ref_g1_to_buffer(p.ctypes.data, buf.ctypes.data)
        return bytes(buf)

    def point_table(self, points):
        p = _arr(points, 8)
        r = np.empty((2 * p.shape[0], 8), dtype=np.uint64)
        self.lib.ref_point_table(p.ctypes.data, p.shape[0], r.ctypes.data)
        return r

    def wnaf_schedule(self, scalars, wnaf_bits):
        """compute_wnaf_states; `wnaf_bits` (= get_optimal_bucket_width(n) + 1) only sizes the output buffer."""
        s = _arr(scalars, 4)
        n = s.shape[0]
        rounds = (127 + wnaf_bits - 1) // wnaf_bits
        sched = np.empty(rounds * 2 * n, dtype=np.uint64)
        skew = np.empty(2 * n, dtype=np.uint8)
        counts = np.zeros(256, dtype=np.uint64)
        wb = self.lib.ref_wnaf_schedule(s.ctypes.data, n, sched.ctypes.data, skew.ctypes.data, counts.ctypes.data)
        assert wb == wnaf_bits, (wb, wnaf_bits)
        return sched.reshape(rounds, 2 * n), skew, counts[:rounds]

    def msm_naive(self, scalars, points):
        s, p = _arr(scalars, 4), _arr(points, 8)
        r = np.empty(8, dtype=np.uint64)
        self.lib.ref_msm_naive(s.ctypes.data, p.ctypes.data, s.shape[0], r.ctypes.data)
        return r

    def g1_sum(self, jacs):
        j = _arr(jacs, 12)
        r = np.empty(8, dtype=np.uint64)
        self.lib.ref_g1_sum(j.ctypes.data, j.shape[0], r.ctypes.data)
        return r

    class Msm:
        def __init__(self, ref, points):
            self.ref = ref
            p = _arr(points, 8)
            self.n = p.shape[0]
            self.h = ref.lib.ref_msm_new(p.ctypes.data, self.n)

        def run(self, scalars, start=0, unsafe=True):
            s = _arr(scalars, 4)
            r = np.empty(8, dtype=np.uint64)
            t = self.ref.lib.ref_msm_run(self.h, s.ctypes.data, start, s.shape[0], 1 if unsafe else 0, r.ctypes.data)
            return r, t

        def run_jac(self, scalars, start=0):
            s = _arr(scalars, 4)
            r = np.empty(12, dtype=np.uint64)
            self.ref.lib.ref_msm_run_jac(self.h, s.ctypes.data, start, s.shape[0], r.ctypes.data)
            return r

        def free(self):
            if self.h:
                self.ref.lib.ref_msm_free(self.h)
                self.h = None

    def msm(self, points): return Ref.Msm(self, points)

    class Domain:
        def __init__(self, ref, log2n, generator_size=0):
            self.ref, self.log2n = ref, log2n
            self.h = ref.lib.ref_domain_new(log2n, generator_size)

        def run(self, coeffs, op=0, constant=None):
            a = _arr(coeffs, 4).copy()
            assert a.shape[0] == 1 << self.log2n
            c = None if constant is None else np.ascontiguousarray(constant, dtype=np.uint64)
            t = self.ref.lib.ref_ntt_run(self.h, a.ctypes.data, op, None if c is None else c.ctypes.data)
            return a, t

        def free(self):
            if self.h:
                self.ref.lib.ref_domain_free(self.h)
                self.h = None

    def domain(self, log2n, generator_size=0): return Ref.Domain(self, log2n, generator_size)

    def coset_fft_split(self, coeffs, ext):
        a = _arr(coeffs, 4)
        n = a.shape[0]
        buf = np.zeros((n * ext, 4), dtype=np.uint64)
        buf[:n] = a
        self.lib.ref_coset_fft_split(buf.ctypes.data, n.bit_length() - 1, ext)
        return buf

    def poly_eval(self, coeffs, z):
        a, z = _arr(coeffs, 4), np.ascontiguousarray(z, dtype=np.uint64)
        r = np.empty(4, dtype=np.uint64)
        self.lib.ref_poly_eval(a.ctypes.data, a.shape[0], z.ctypes.data, r.ctypes.data)
        return r

    def poly_binop(self, op, a, b):
        a, b = _arr(a, 4), _arr(b, 4)
        r = np.empty_like(a)
        self.lib.ref_poly_binop(op, a.ctypes.data, b.ctypes.data, r.ctypes.data, a.shape[0].bit_length() - 1)
        return r

    def kate_opening(self, src, z):
        a, z = _arr(src, 4), np.ascontiguousarray(z, dtype=np.uint64)
        dest, f = np.empty_like(a), np.empty(4, dtype=np.uint64)
        self.lib.ref_kate_opening(a.ctypes.data, dest.ctypes.data, a.shape[0], z.ctypes.data, f.ctypes.data)
        return dest, f

    def divide_by_pseudo_vanishing(self, evals, log2_src, cut=4):
        a = _arr(evals, 4).copy()
        self.lib.ref_divide_by_pseudo_vanishing(a.ctypes.data, log2_src, a.shape[0].bit_length() - 1, cut)
        return a


# ------------------------------------------------------------------------------------------------------------------
# The reference's whole TurboPLONK prover with its MSM / FFT work items delegated to callbacks (ref_prover_driver.cpp)
PROVER_SO = os.path.join(_HERE, "_ref", "libbbprover.so")
# same objects + shim/bbg_barretenberg_shim.cpp + -Wl,--wrap flags + libbbg.so: the reference prover with its MSM / FFT
# entry points wrapped onto the GPU library at link time.  Needs the HIP runtime: import torch first.
PROVER_GPU_SO = os.path.join(_HERE, "_ref", "libbbprover_gpu.so")
# the CPU build's own driver object linked with BOTH shim TUs: construct_proof() itself is wrapped (oracle/Makefile prover_wrap)
PROVER_WRAP_SO = os.environ.get("BBG_PROVER_WRAP_SO") or os.path.join(_HERE, "_ref", "libbbprover_wrap.so")  # env: the sanitizer build (scripts/sanitize_run.sh)


def prover_available():
    if not os.path.exists(PROVER_SO):
        return False
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return all(f in flags for f in (" adx", " bmi2", " avx2"))


class RefProver:
    """One TurboComposer circuit + TurboProver of the reference.  `engine` supplies the three hot-path operations:
         engine.msm(scalars[n,4]) -> jacobian[12]        (over the monomials returned by .monomials())
         engine.coset_fft(coeffs[4n,4], generator_size)  -> coeffs[4n,4]
         engine.ifft(coeffs[n,4])                        -> coeffs[n,4]
       prove(engine=None) uses the reference's own CPU process_queue."""

    MSM_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p)
    FFT_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p)
    IFFT_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
    FFT_ITEM_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
    OPENING_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_void_p)

    def __init__(self, num_gates, circuit_seed, points, x_mont, gpu_linked=False, flavour=0, wrap_linked=False):
        """flavour 0 = TurboPLONK (TurboComposer / TurboProver), 1 = StandardPLONK (StandardComposer / Prover) over the same circuit,
        2 = MiMCComposer (MiMC rounds + the arithmetic chain; Prover with the MiMC widget).
        gpu_linked: libbbprover_gpu.so (MSM / FFT entry points wrapped + the explicit resident glue); wrap_linked: libbbprover_wrap.so (the
        CPU build's driver object, construct_proof() itself wrapped: prove_reference() IS the resident prover there)."""
        so = PROVER_WRAP_SO if wrap_linked else PROVER_GPU_SO if gpu_linked else PROVER_SO
        if not prover_available() or not os.path.exists(so):
            raise RuntimeError(f"{so} not available on this machine")
        L = self.lib = ctypes.CDLL(so, mode=os.RTLD_NOW)
        L.refp_gpu_linked.restype = cint
        L.refp_wrap_linked.restype = cint
        assert bool(L.refp_gpu_linked()) == bool(gpu_linked or wrap_linked) and bool(L.refp_wrap_linked()) == bool(wrap_linked)
        L.refp_wrap_set_enabled.argtypes = [cint]
        L.refp_wrap_set_replay.argtypes = [vp, sz]
        L.refp_wrap_set_budget.argtypes = [sz]
        L.refp_wrap_cached_keys.restype = sz
        L.refp_wrap_bytes.restype = sz
        L.refp_wrap_trim.restype = sz
        L.refp_wrap_stats.argtypes = [vp]
        L.refp_wrap_reuploads.restype = ctypes.c_uint64
        L.refp_wrap_in_progress.restype = sz
        L.refp_construct_proof_rounds.argtypes = [vp, vp, vp]; L.refp_construct_proof_rounds.restype = ctypes.c_double
        L.refp_wrap_fail_round.argtypes = [cint]; L.refp_wrap_fail_round.restype = cint
        L.refp_shim_option.argtypes = [ctypes.c_char_p, ctypes.c_long]; L.refp_shim_option.restype = cint
        L.refp_key_selector_scale3.argtypes = [vp, ctypes.c_char_p]; L.refp_key_selector_scale3.restype = cint
        if hasattr(L, "refp_key_selector_poke"):
            L.refp_key_selector_poke.argtypes = [vp, ctypes.c_char_p, sz]; L.refp_key_selector_poke.restype = cint
        L.refp_reset.argtypes = [vp]
        L.refp_new_flavour.argtypes = [cint, sz, ctypes.c_uint64, vp, sz, vp]; L.refp_new_flavour.restype = vp
        L.refp_program_width.argtypes = [vp]; L.refp_program_width.restype = sz
        L.refp_construct_proof_recording.argtypes = [vp, vp]; L.refp_construct_proof_recording.restype = cint
        L.refp_construct_proof_reference.argtypes = [vp]; L.refp_construct_proof_reference.restype = cint
        L.refp_resident_key_create.argtypes = [vp]; L.refp_resident_key_create.restype = ctypes.c_double
        L.refp_construct_proof_resident.argtypes = [vp, vp, sz]; L.refp_construct_proof_resident.restype = ctypes.c_double
        L.refp_last_error.argtypes = [vp]; L.refp_last_error.restype = ctypes.c_char_p
        L.refio_read_transcript_g1.argtypes = [ctypes.c_char_p, sz, vp]; L.refio_read_transcript_g1.restype = cint
        L.refp_circuit_size.argtypes = [vp]; L.refp_circuit_size.restype = sz
        L.refp_get_monomials.argtypes = [vp, vp, sz]
        L.refp_execute_round.argtypes = [vp, cint]; L.refp_execute_round.restype = sz
        L.refp_process_queue_reference.argtypes = [vp]
        L.refp_process_queue_with2.argtypes = [vp, self.MSM_CB, self.FFT_CB, self.FFT_ITEM_CB, self.IFFT_CB, vp, cint, vp]
        L.refp_process_queue_with2.restype = cint
        L.refp_export_proof.argtypes = [vp, vp, sz]; L.refp_export_proof.restype = sz
        L.refp_verify.argtypes = [vp]; L.refp_verify.restype = cint
        L.refp_delete.argtypes = [vp]
        L.refp_set_threads.argtypes = [cint]
        L.refp_max_threads.restype = cint
        # small circuits: cap the reference's OpenMP team (see refp_set_threads in ref_prover_driver.cpp)
        self.threads = min(os.cpu_count() or 1, 8 if num_gates < (1 << 15) else 16 if num_gates < (1 << 17) else 64)
        L.refp_set_threads(self.threads)
        pts = _arr(points, 8)
        x = np.ascontiguousarray(x_mont, dtype=np.uint64)
        self.h = L.refp_new_flavour(flavour, num_gates, circuit_seed, pts.ctypes.data, pts.shape[0], x.ctypes.data)
        if not self.h:
            raise RuntimeError("refp_new failed (circuit larger than the SRS?)")
        self.n = int(L.refp_circuit_size(self.h))
        self.counts = [0, 0, 0]
        self.mismatches = 0

    # ---- whole proofs
    def _proof(self):
        size = self.lib.refp_export_proof(self.h, None, 0)
        buf = (ctypes.c_uint8 * size)()
        self.lib.refp_export_proof(self.h, buf, size)
        return bytes(buf)

    def prove_recording(self):
        """ProverBase::construct_proof round by round, recording the blinding scalars it draws: (proof bytes, (3w + 3, 4) limbs)."""
        w = int(self.lib.refp_program_width(self.h))
        blind = np.zeros((3 * w + 3, 4), dtype=np.uint64)
        rc = self.lib.refp_construct_proof_recording(self.h, blind.ctypes.data)
        if rc != 3 * w + 3:
            raise RuntimeError(f"refp_construct_proof_recording failed ({rc})")
        return self._proof(), blind

    def prove_reference(self, replay=None, reset=False):
        """ProverBase::construct_proof() as shipped, one call (shim-linked build: MSM / FFT entry points on the GPU; wrap-linked build: the
        resident prover behind the wrapped symbol).  replay (wrap-linked only): blinding scalars for this proof; reset: ProverBase::reset()
        first (a second proof on the same prover object)."""
        if reset:
            self.lib.refp_reset(self.h)
        if replay is not None:
            r = _arr(replay, 4)
            self.lib.refp_wrap_set_replay(r.ctypes.data, r.shape[0])
        try:
            if self.lib.refp_construct_proof_reference(self.h) != 0:
                raise RuntimeError("refp_construct_proof_reference failed: " + (self.lib.refp_last_error(self.h) or b"").decode())
        finally:
            if replay is not None:
                self.lib.refp_wrap_set_replay(None, 0)
        return self._proof()

    # ---- the wrapped construct_proof() (wrap-linked build): controls and counters of shim/bbg_prover_wrap.cpp
    def wrap_set_enabled(self, on):
        self.lib.refp_wrap_set_enabled(1 if on else 0)

    def wrap_set_budget(self, nbytes):
        self.lib.refp_wrap_set_budget(nbytes)

    def wrap_cached_keys(self):
        return int(self.lib.refp_wrap_cached_keys())

    def wrap_bytes(self):
        return int(self.lib.refp_wrap_bytes())

    def wrap_trim(self):
        return int(self.lib.refp_wrap_trim())

    def wrap_clear(self):
        self.lib.refp_wrap_clear()

    def wrap_stats(self):
        """(proofs through the resident path, fallbacks to the reference body, evictions)."""
        out = (ctypes.c_uint64 * 3)()
        self.lib.refp_wrap_stats(out)
        return tuple(int(v) for v in out)

    def wrap_in_progress(self):
        """Proofs the wrap holds in progress round by round."""
        return int(self.lib.refp_wrap_in_progress())

    def prove_round_by_round(self, replay=None, reset=False, order=None):
        """The seven execute_*_round entry points + process_queue() between them + export_proof, the way a host of the reference's C binding
        drives a prover (plonk/proof_system/prover/c_bind.cpp:59-92).  Returns (proof bytes, seconds, work items queued after each round).
        In the wrap-linked build the rounds are the wrapped symbols (shim/bbg_prover_wrap.cpp); replay / reset as prove_reference."""
        if reset:
            self.lib.refp_reset(self.h)
        if replay is not None:
            r = _arr(replay, 4)
            self.lib.refp_wrap_set_replay(r.ctypes.data, r.shape[0])
        q = (ctypes.c_size_t * 7)()
        o = (ctypes.c_int * 7)(*order) if order is not None else None
        try:
            t = self.lib.refp_construct_proof_rounds(self.h, o, q)
            if t < 0:
                raise RuntimeError("refp_construct_proof_rounds failed: " + (self.lib.refp_last_error(self.h) or b"").decode())
        finally:
            if replay is not None:
                self.lib.refp_wrap_set_replay(None, 0)
        size = self.lib.refp_export_proof(self.h, None, 0)
        buf = (ctypes.c_uint8 * size)()
        self.lib.refp_export_proof(self.h, buf, size)
        return bytes(buf), float(t), [int(v) for v in q]

    def wrap_reuploads(self):
        """Keys uploaded again because a cached proving key's host polynomials had changed."""
        return int(self.lib.refp_wrap_reuploads())

    def shim_option(self, key, value):
        """bbg_set_option on the context the linked shim proves with"""
        if self.lib.refp_shim_option(key.encode(), int(value)) != 0:
            raise RuntimeError(f"refp_shim_option({key}): refused, or not a shim-linked build")

    def wrap_fail_round(self, rnd):
        """The next resident prover round `rnd` fails once (library option prover_fail_round, tests only)."""
        if self.lib.refp_wrap_fail_round(rnd) != 0:
            raise RuntimeError("refp_wrap_fail_round: not a wrap-linked build")

    def key_selector_scale3(self, label):
        """Rewrites selector `label` of this session's proving key in place (coefficients * 3, 4n coset form recomputed)."""
        if self.lib.refp_key_selector_scale3(self.h, label.encode()) != 0:
            raise RuntimeError("refp_key_selector_scale3 failed")

    def key_selector_poke(self, label, index):
        """ONE coefficient of selector `label` of this session's proving key += 1 (4n coset form recomputed)."""
        if self.lib.refp_key_selector_poke(self.h, label.encode(), int(index)) != 0:
            raise RuntimeError("refp_key_selector_poke failed")

    def resident_key_create(self):
        """bbg_shim::ResidentKey for this circuit's proving key (shim-linked build only); seconds."""
        t = self.lib.refp_resident_key_create(self.h)
        if t < 0:
            raise RuntimeError("resident key: " + (self.lib.refp_last_error(self.h) or b"not a shim-linked build").decode())
        return t

    def resident_check_key(self):
        """Number of key polynomials whose device-derived form (sigma Lagrange, 4n coset, L_1) differs from the reference's own array."""
        self.lib.refp_resident_check_key.argtypes = [vp]; self.lib.refp_resident_check_key.restype = cint
        rc = self.lib.refp_resident_check_key(self.h)
        if rc < 0:
            raise RuntimeError("resident key check: " + (self.lib.refp_last_error(self.h) or b"not a shim-linked build").decode())
        return rc

    def prove_resident(self, replay=None):
        """bbg_shim::construct_proof (shim/bbg_resident_prover.hpp): every O(n) step on the device.  replay: blinding scalars to
        use instead of fr::random_element() (what prove_recording returned).  Returns (proof bytes, seconds)."""
        if replay is None:
            t = self.lib.refp_construct_proof_resident(self.h, None, 0)
        else:
            r = _arr(replay, 4)
            t = self.lib.refp_construct_proof_resident(self.h, r.ctypes.data, r.shape[0])
        if t < 0:
            raise RuntimeError("resident proof: " + (self.lib.refp_last_error(self.h) or b"not a shim-linked build").decode())
        return self._proof(), t

    def read_transcript_g1(self, directory, degree):
        """The reference's own io::read_transcript_g1 (srs/io.cpp:134-162)."""
        out = np.zeros((degree, 8), dtype=np.uint64)
        if self.lib.refio_read_transcript_g1(str(directory).encode(), degree, out.ctypes.data) != 0:
            raise RuntimeError("io::read_transcript_g1 threw")
        return out

    def monomials(self, count=None):
        count = self.n + 1 if count is None else count
        out = np.empty((count, 8), dtype=np.uint64)
        self.lib.refp_get_monomials(self.h, out.ctypes.data, count)
        return out

    def prove(self, engine=None, check=True):
        """Runs all rounds (prover.cpp:420-436 construct_proof); returns the proof bytes."""
        cbs = None
        if engine is not None and getattr(engine, "queue_via_reference", False):
            pass  # the prover's own process_queue handles the items (shim-linked build); the engine only takes round 4
        elif engine is not None and getattr(engine, "raw", False):
            # raw engine: gets the prover's own buffers (addresses) and works in place -- what a C++ binding does
            item = None
            if hasattr(engine, "fft_item_raw"):  # the whole FFT work item (n coefficients -> 4n + 4 values) in one call
                item = self.FFT_ITEM_CB(lambda w, lgn, wf, lgd, _u: engine.fft_item_raw(w, lgn, wf, lgd))
            cbs = (self.MSM_CB(lambda scalars, n, out, _u: engine.msm_raw(scalars, n, out)),
                   self.FFT_CB(lambda coeffs, lg, gs, _u: engine.coset_fft_raw(coeffs, lg, gs)),
                   self.IFFT_CB(lambda coeffs, lg, _u: engine.ifft_raw(coeffs, lg)), item)
        elif engine is not None:
            def msm_cb(scalars, n, out, _user):
                s = np.ctypeslib.as_array(ctypes.cast(scalars, ctypes.POINTER(ctypes.c_uint64)), shape=(n, 4))
                r = np.ascontiguousarray(engine.msm(s), dtype=np.uint64)
                ctypes.memmove(out, r.ctypes.data, 96)

            def fft_cb(coeffs, log2_domain, generator_size, _user):
                m = 1 << log2_domain
                a = np.ctypeslib.as_array(ctypes.cast(coeffs, ctypes.POINTER(ctypes.c_uint64)), shape=(m, 4))
                r = np.ascontiguousarray(engine.coset_fft(a, generator_size), dtype=np.uint64)
                ctypes.memmove(coeffs, r.ctypes.data, m * 32)

            def ifft_cb(coeffs, log2n, _user):
                m = 1 << log2n
                a = np.ctypeslib.as_array(ctypes.cast(coeffs, ctypes.POINTER(ctypes.c_uint64)), shape=(m, 4))
                r = np.ascontiguousarray(engine.ifft(a), dtype=np.uint64)
                ctypes.memmove(coeffs, r.ctypes.data, m * 32)

            item = None
            if hasattr(engine, "fft_item"):
                def item_cb(wire, log2n, wire_fft, log2_domain, _user):
                    a = np.ctypeslib.as_array(ctypes.cast(wire, ctypes.POINTER(ctypes.c_uint64)), shape=(1 << log2n, 4))
                    r = np.ascontiguousarray(engine.fft_item(a, log2_domain), dtype=np.uint64)
                    assert r.shape == ((1 << log2_domain) + 4, 4)
                    ctypes.memmove(wire_fft, r.ctypes.data, r.nbytes)
                item = self.FFT_ITEM_CB(item_cb)
            cbs = (self.MSM_CB(msm_cb), self.FFT_CB(fft_cb), self.IFFT_CB(ifft_cb), item)
        self.counts = [0, 0, 0]
        self.mismatches = 0
        self.t_rounds = self.t_queue = 0.0  # seconds in the prover's own round logic / in the MSM+FFT work items
        self.t_round = [0.0] * 7           # per execute_*_round (0 = preamble)
        import time
        self.round4_mismatch = None
        for k in range(7):
            t0 = time.perf_counter()
            if k == 3 and engine is not None and hasattr(engine, "round3_raw"):
                self._round3_with(engine)
            elif k == 4 and engine is not None and hasattr(engine, "round4_raw"):
                self._round4_with(engine, check)
            elif k == 6 and engine is not None and hasattr(engine, "round6_raw"):
                self._round6_with(engine)
            else:
                self.lib.refp_execute_round(self.h, k)
            t1 = time.perf_counter()
            self.t_rounds += t1 - t0
            self.t_round[k] = t1 - t0
            if k == 5:
                continue  # construct_proof() runs rounds 5 and 6 back to back
            if cbs is None:
                self.lib.refp_process_queue_reference(self.h)
                self.t_queue += time.perf_counter() - t1
            else:
                c = (ctypes.c_uint32 * 3)()
                item = cbs[3] if cbs[3] is not None else ctypes.cast(None, self.FFT_ITEM_CB)
                rc = self.lib.refp_process_queue_with2(self.h, cbs[0], cbs[1], item, cbs[2], None, 1 if check else 0, c)
                if rc < 0:
                    raise RuntimeError(f"refp_process_queue_with failed ({rc}) in round {k}")
                self.mismatches += rc
                for i in range(3):
                    self.counts[i] += c[i]
                self.t_queue += time.perf_counter() - t1
        size = self.lib.refp_export_proof(self.h, None, 0)
        buf = (ctypes.c_uint8 * size)()
        self.lib.refp_export_proof(self.h, buf, size)
        return bytes(buf)

    def _round3_with(self, engine):
        """execute_third_round with the permutation polynomial z (grand product, blinding, ifft) computed by the engine."""
        L = self.lib
        L.refp_round3_begin.argtypes = [vp, vp, vp, vp, vp, vp]; L.refp_round3_begin.restype = cint
        L.refp_round3_end.argtypes = [vp]; L.refp_round3_end.restype = cint
        wires = (ctypes.c_void_p * 4)()
        sigmas = (ctypes.c_void_p * 4)()
        ch = np.zeros((5, 4), dtype=np.uint64)
        blind = np.zeros((3, 4), dtype=np.uint64)
        z = ctypes.c_void_p()
        log2n = L.refp_round3_begin(self.h, wires, sigmas, ch.ctypes.data, blind.ctypes.data, ctypes.byref(z))
        if log2n < 0:
            raise RuntimeError("refp_round3_begin failed")
        engine.round3_raw([int(p) for p in wires], [int(p) for p in sigmas], ch, blind, log2n, z.value)
        if L.refp_round3_end(self.h) != 0:
            raise RuntimeError("refp_round3_end failed")

    def _round6_with(self, engine):
        """execute_sixth_round with the opening polynomials (accumulation + Kate division) computed by the engine."""
        L = self.lib
        L.refp_round6_with.argtypes = [vp, self.OPENING_CB, vp]; L.refp_round6_with.restype = cint

        def cb(pz, sz_, cz, base, po, so, co, zeta, zeta_omega, n, wz, wzo, _user):
            def ptrs(arr, cnt):
                return [int(v) for v in ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p * cnt)).contents] if cnt else []

            def limbs(ptr, cnt):
                return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint64)), shape=(cnt, 4)).copy()
            engine.round6_raw(ptrs(pz, cz), limbs(sz_, cz), base, ptrs(po, co), limbs(so, co), limbs(zeta, 1)[0], limbs(zeta_omega, 1)[0],
                              n, wz, wzo)

        keep = self.OPENING_CB(cb)
        if L.refp_round6_with(self.h, keep, None) != 0:
            raise RuntimeError("refp_round6_with failed")

    def _round4_with(self, engine, check):
        """execute_fourth_round with the quotient (widgets + divide_by_pseudo_vanishing + coset_ifft) computed by the engine."""
        L = self.lib
        L.refp_round4_begin.argtypes = [vp, vp, vp, vp]; L.refp_round4_begin.restype = cint
        L.refp_round4_reference_quotient.argtypes = [vp]; L.refp_round4_reference_quotient.restype = cint
        L.refp_round4_end.argtypes = [vp]; L.refp_round4_end.restype = cint
        ptrs = (ctypes.c_void_p * 21)()
        ch = np.zeros((9, 4), dtype=np.uint64)
        q = ctypes.c_void_p()
        log2n = L.refp_round4_begin(self.h, ptrs, ch.ctypes.data, ctypes.byref(q))
        if log2n < 0:
            raise RuntimeError("refp_round4_begin failed")
        m = 4 << log2n
        engine.round4_raw([int(p) for p in ptrs], ch, log2n, q.value)
        if check:
            got = np.ctypeslib.as_array(ctypes.cast(q.value, ctypes.POINTER(ctypes.c_uint64)), shape=(m, 4)).copy()
            assert L.refp_round4_reference_quotient(self.h) == 0
            want = np.ctypeslib.as_array(ctypes.cast(q.value, ctypes.POINTER(ctypes.c_uint64)), shape=(m, 4))
            from_canon = Oracle().canon
            self.round4_mismatch = int(not np.array_equal(from_canon(0, got), from_canon(0, want)))
            self.mismatches += self.round4_mismatch
        if L.refp_round4_end(self.h) != 0:
            raise RuntimeError("refp_round4_end failed")

    def verify(self):
        return int(self.lib.refp_verify(self.h))

    def free(self):
        if self.h:
            self.lib.refp_delete(self.h)
            self.h = None


class RefWidgets:
    """The reference's own quotient widgets of a TurboProver (permutation, turbo arithmetic / fixed base / range / logic) run
    one at a time over caller-supplied "*_fft" arrays with a deterministic transcript (ref_prover_driver.cpp, refw_*)."""

    # label of each polynomial in the order of include/bbg.h's bbg_quotient_poly
    LABELS = ["w_1_fft", "w_2_fft", "w_3_fft", "w_4_fft", "z_fft", "sigma_1_fft", "sigma_2_fft", "sigma_3_fft", "sigma_4_fft",
              "q_1_fft", "q_2_fft", "q_3_fft", "q_4_fft", "q_5_fft", "q_m_fft", "q_c_fft", "q_arith_fft", "q_ecc_1_fft",
              "q_range_fft", "q_logic_fft", "lagrange_1"]
    MIMC_LABELS = LABELS + ["q_mimc_coefficient_fft", "q_mimc_selector_fft"]  # include/bbg.h's extended table (BBG_QP_EXT_*)

    def __init__(self, prover, standard=None, flavour=None):
        """prover: a RefProver (TurboPLONK widgets 0..4).  standard=(num_gates, points, x_mont): instead build a StandardPLONK key
        through the same library; its widgets are 0 = permutation over three wires, 1 = arithmetic."""
        self.prover = prover  # keeps the session (proving key) alive
        L = self.lib = prover.lib
        L.refw_new.argtypes = [vp]; L.refw_new.restype = vp
        L.refw_delete.argtypes = [vp]
        L.refw_poly_size.argtypes = [vp, ctypes.c_char_p]; L.refw_poly_size.restype = sz
        L.refw_set_poly.argtypes = [vp, ctypes.c_char_p, vp, sz]; L.refw_set_poly.restype = cint
        L.refw_get_poly.argtypes = [vp, ctypes.c_char_p, vp, sz]; L.refw_get_poly.restype = cint
        L.refw_challenges.argtypes = [vp, vp]
        L.refw_run_widget.argtypes = [vp, cint, vp, vp]; L.refw_run_widget.restype = cint
        L.refw_new_standard.argtypes = [sz, vp, sz, vp]; L.refw_new_standard.restype = vp
        L.refw_circuit_size.argtypes = [vp]; L.refw_circuit_size.restype = sz
        L.refw_new_flavour.argtypes = [vp, cint]; L.refw_new_flavour.restype = vp
        if flavour is not None:  # the widget objects of `prover`'s own session (RefProver(flavour=...)): 2 = permutation<3>, MiMC, arithmetic
            self.h = L.refw_new_flavour(prover.h, flavour)
        elif standard is None:
            self.h = L.refw_new(prover.h)
        else:
            gates, pts, x = standard
            pts = _arr(pts, 8)
            x = np.ascontiguousarray(x, dtype=np.uint64)
            self.h = L.refw_new_standard(gates, pts.ctypes.data, pts.shape[0], x.ctypes.data)
        if not self.h:
            raise RuntimeError("refw_new failed")
        self.m = 4 * int(L.refw_circuit_size(self.h))  # large (coset) domain size

    def has_poly(self, label):
        return int(self.lib.refw_poly_size(self.h, label.encode())) > 0

    def set_poly(self, label, data):
        a = _arr(data, 4)
        assert self.lib.refw_set_poly(self.h, label.encode(), a.ctypes.data, a.shape[0]) == 0, label

    def get_poly(self, label, count=None):
        count = self.m if count is None else count
        out = np.empty((count, 4), dtype=np.uint64)
        assert self.lib.refw_get_poly(self.h, label.encode(), out.ctypes.data, count) == 0, label
        return out

    def challenges(self):
        """(8, 4): alpha, beta, gamma, public_input_delta, k1, k2, k3, g  (Montgomery)."""
        out = np.zeros((8, 4), dtype=np.uint64)
        self.lib.refw_challenges(self.h, out.ctypes.data)
        return out

    def run(self, widget, alpha_base):
        a = np.ascontiguousarray(alpha_base, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        assert self.lib.refw_run_widget(self.h, widget, a.ctypes.data, out.ctypes.data) == 0
        return out

    def free(self):
        if self.h:
            self.lib.refw_delete(self.h)
            self.h = None
