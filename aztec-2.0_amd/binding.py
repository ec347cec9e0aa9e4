"""ctypes binding of include/bbg.h (libbbg.so).  Host arrays are numpy uint64 in the reference's layout:
scalars / coefficients (n, 4); affine points (n, 8); Jacobian points (n, 12) or (12,)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("BBG_LIB_PATH") or os.path.join(CSRC, "libbbg.so")  # BBG_LIB_PATH: A/B builds of the same ABI (tests/tools), never a fallback

# op codes of bbg_ntt (include/bbg.h)
FFT, IFFT, COSET_FFT, COSET_IFFT = 0, 1, 2, 3
FFT_WITH_CONSTANT, COSET_FFT_WITH_CONSTANT, COSET_FFT_WITH_GENERATOR_SHIFT, IFFT_WITH_CONSTANT = 4, 5, 6, 7


class BbgError(RuntimeError):
    pass


def build_library(force=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC, "-j8"], check=True)
    return LIB_PATH


_lib = None


def load_library():
    """dlopen libbbg.so and declare prototypes.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BbgError(f"{LIB_PATH} is missing: run __graft_entry__.build() (there is no CPU fallback)")
    # One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64 (soname libamdhip64.so.7, the
    # same soname libbbg.so needs).  If torch is imported AFTER libbbg pulled in /opt/rocm's copy, torch loads a
    # second runtime by file name and finds "No HIP GPUs".  Importing torch first makes the loader resolve
    # libbbg's NEEDED entry to the copy torch already mapped, so both share devices, streams and allocations.
    if os.environ.get("BBG_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, sz, u64p, cint = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int
    protos = {
        "bbg_device_count": (cint, []),
        "bbg_init": (cint, [cint, ctypes.POINTER(vp)]),
        "bbg_destroy": (None, [vp]),
        "bbg_last_error": (ctypes.c_char_p, []),
        "bbg_sync": (cint, [vp]),
        "bbg_join": (cint, [vp]),
        "bbg_join_lag": (cint, [vp, cint]),
        "bbg_set_stream": (cint, [vp, vp]),
        "bbg_srs_register": (cint, [vp, vp, sz, sz, ctypes.POINTER(vp)]),
        "bbg_srs_register_device": (cint, [vp, vp, sz, ctypes.POINTER(vp)]),
        "bbg_srs_synth_linear": (cint, [vp, ctypes.c_uint64, ctypes.c_uint64, sz, ctypes.POINTER(vp)]),
        "bbg_srs_synth_hashed": (cint, [vp, ctypes.c_uint64, sz, ctypes.POINTER(vp)]),
        "bbg_srs_load_transcript": (cint, [vp, ctypes.c_char_p, sz, ctypes.POINTER(vp)]),
        "bbg_srs_register_transcript_buffer": (cint, [vp, vp, sz, ctypes.POINTER(vp)]),
        "bbg_srs_write_transcript": (cint, [vp, ctypes.c_char_p, sz, vp]),
        "bbg_transcript_checksum": (cint, [vp, sz, vp]),
        "bbg_srs_num_points": (sz, [vp]),
        "bbg_srs_read": (cint, [vp, sz, sz, vp]),
        "bbg_srs_free": (None, [vp]),
        "bbg_srs_retain": (cint, [vp]),
        "bbg_msm": (cint, [vp, vp, vp, sz, sz, vp]),
        "bbg_msm_device": (cint, [vp, vp, vp, sz, sz, vp]),
        "bbg_msm_batch": (cint, [vp, vp, sz, vp, vp, vp, vp]),
        "bbg_msm_batch_device": (cint, [vp, vp, sz, vp, vp, vp, vp]),
        "bbg_msm_plan": (cint, [vp, vp, sz, ctypes.POINTER(cint), ctypes.POINTER(cint)]),
        "bbg_ntt_plan": (cint, [vp, ctypes.c_uint, ctypes.POINTER(cint), ctypes.POINTER(cint), ctypes.POINTER(cint), ctypes.POINTER(cint)]),
        "bbg_memory_report": (cint, [vp, vp]),
        "bbg_memory_trim": (cint, [vp, cint, ctypes.POINTER(sz)]),
        "bbg_prover_device_bytes": (cint, [vp, ctypes.POINTER(sz)]),
        "bbg_g1_sum": (cint, [vp, vp, sz, vp]),
        "bbg_g1_sum_device": (cint, [vp, vp, sz, vp]),
        "bbg_g1_normalize": (cint, [vp, vp, sz, vp]),
        "bbg_ntt": (cint, [vp, vp, ctypes.c_uint, cint, sz, vp]),
        "bbg_coset_fft_extend": (cint, [vp, vp, ctypes.c_uint, ctypes.c_uint, vp]),
        "bbg_quotient_widget_device": (cint, [vp, cint, vp, ctypes.c_uint, vp, vp, vp]),
        "bbg_poly_linear_combination_device": (cint, [vp, vp, vp, sz, vp, vp, sz]),
        "bbg_permutation_grand_product_device": (cint, [vp, vp, vp, ctypes.c_uint, vp, vp]),
        "bbg_poly_evaluate": (cint, [vp, vp, sz, vp, vp]),
        "bbg_kate_opening": (cint, [vp, vp, vp, sz, vp, vp]),
        "bbg_divide_by_pseudo_vanishing": (cint, [vp, vp, ctypes.c_uint, ctypes.c_uint, sz]),
        "bbg_ntt_device": (cint, [vp, vp, ctypes.c_uint, cint, sz, vp]),
        "bbg_ntt_prepare": (cint, [vp, ctypes.c_uint]),
        "bbg_coset_fft_split": (cint, [vp, vp, ctypes.c_uint, sz]),
        "bbg_coset_fft_split_device": (cint, [vp, vp, ctypes.c_uint, sz]),
        "bbg_scale_powers_device": (cint, [vp, vp, sz, vp, vp]),
        "bbg_fr_root_pow": (cint, [vp, ctypes.c_uint, ctypes.c_uint64, cint, vp]),
        "bbg_fr_pow": (cint, [vp, vp, ctypes.c_uint64, vp]),
        "bbg_cross_dft_device": (cint, [vp, vp, vp, ctypes.c_uint, sz, ctypes.c_uint, cint]),
        "bbg_poly_op_device": (cint, [vp, cint, vp, vp, vp, sz]),
        "bbg_poly_evaluate_device": (cint, [vp, vp, sz, vp, vp]),
        "bbg_kate_opening_device": (cint, [vp, vp, vp, sz, vp, vp]),
        "bbg_divide_by_pseudo_vanishing_device": (cint, [vp, vp, ctypes.c_uint, ctypes.c_uint, sz]),
        "bbg_dev_alloc": (cint, [vp, sz, ctypes.POINTER(vp)]),
        "bbg_dev_free": (cint, [vp, vp]),
        "bbg_dev_upload": (cint, [vp, vp, vp, sz]),
        "bbg_dev_download": (cint, [vp, vp, vp, sz]),
        "bbg_set_option": (cint, [vp, ctypes.c_char_p, ctypes.c_long]),
        "bbg_field_op": (cint, [vp, cint, cint, vp, vp, vp, sz]),
        "bbg_profile_enable": (cint, [vp, cint]),
        "bbg_profile_get": (cint, [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(sz)]),
        "bbg_prover_create": (cint, [vp, vp, ctypes.c_uint, cint, vp, ctypes.POINTER(vp)]),
        "bbg_prover_create_flavour": (cint, [vp, vp, ctypes.c_uint, cint, vp, ctypes.POINTER(vp)]),
        "bbg_prover_destroy": (None, [vp]),
        "bbg_prover_set_key_poly": (cint, [vp, cint, cint, vp]),
        "bbg_prover_finalize_key": (cint, [vp]),
        "bbg_prover_round1": (cint, [vp, vp, vp]),
        "bbg_prover_round3": (cint, [vp, vp, vp, vp, vp]),
        "bbg_prover_round4": (cint, [vp, vp, vp, vp]),
        "bbg_prover_evaluate": (cint, [vp, sz, vp, vp, vp, vp]),
        "bbg_prover_linearise": (cint, [vp, sz, vp, vp, vp, vp]),
        "bbg_prover_round6": (cint, [vp, sz, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp]),
        "bbg_prover_read_poly": (cint, [vp, cint, cint, vp, sz]),
        "bbg_multi_create": (cint, [vp, cint, ctypes.POINTER(vp)]),
        "bbg_multi_destroy": (None, [vp]),
        "bbg_multi_count": (cint, [vp]),
        "bbg_multi_ctx": (vp, [vp, cint]),
        "bbg_multi_sync": (cint, [vp]),
        "bbg_multi_srs_register": (cint, [vp, vp, sz, sz]),
        "bbg_multi_srs_synth_hashed": (cint, [vp, ctypes.c_uint64, sz]),
        "bbg_multi_srs_num_points": (sz, [vp]),
        "bbg_multi_msm": (cint, [vp, vp, sz, sz, vp]),
        "bbg_multi_set_option": (cint, [vp, ctypes.c_char_p, ctypes.c_long]),
        "bbg_multi_ntt_device": (cint, [vp, vp, ctypes.c_uint, cint]),
        "bbg_multi_ntt": (cint, [vp, vp, ctypes.c_uint, cint]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)  # AttributeError here == a symbol declared in bbg.h is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "bbg_device_count", "bbg_init", "bbg_destroy", "bbg_last_error", "bbg_sync", "bbg_join", "bbg_join_lag", "bbg_set_stream", "bbg_srs_register",
    "bbg_srs_register_device", "bbg_srs_synth_linear", "bbg_srs_synth_hashed", "bbg_srs_load_transcript", "bbg_srs_num_points", "bbg_srs_read",
    "bbg_srs_free", "bbg_srs_retain", "bbg_msm", "bbg_msm_device", "bbg_msm_batch", "bbg_msm_batch_device", "bbg_msm_plan", "bbg_ntt_plan", "bbg_memory_report",
    "bbg_memory_trim", "bbg_prover_device_bytes", "bbg_g1_sum", "bbg_g1_sum_device", "bbg_g1_normalize", "bbg_ntt", "bbg_ntt_device", "bbg_coset_fft_extend", "bbg_quotient_widget_device", "bbg_poly_linear_combination_device", "bbg_permutation_grand_product_device", "bbg_poly_evaluate", "bbg_kate_opening",
    "bbg_divide_by_pseudo_vanishing",
    "bbg_ntt_prepare", "bbg_coset_fft_split", "bbg_coset_fft_split_device", "bbg_scale_powers_device", "bbg_fr_root_pow", "bbg_fr_pow", "bbg_cross_dft_device", "bbg_poly_op_device", "bbg_poly_evaluate_device", "bbg_kate_opening_device",
    "bbg_divide_by_pseudo_vanishing_device", "bbg_dev_alloc", "bbg_dev_free",
    "bbg_dev_upload", "bbg_dev_download", "bbg_set_option", "bbg_field_op", "bbg_profile_enable", "bbg_profile_get",
    "bbg_srs_write_transcript", "bbg_transcript_checksum", "bbg_srs_register_transcript_buffer",
    "bbg_prover_create", "bbg_prover_create_flavour", "bbg_prover_destroy", "bbg_prover_set_key_poly", "bbg_prover_finalize_key", "bbg_prover_round1", "bbg_prover_round3",
    "bbg_prover_round4", "bbg_prover_evaluate", "bbg_prover_linearise", "bbg_prover_round6", "bbg_prover_read_poly",
    "bbg_multi_create", "bbg_multi_destroy", "bbg_multi_count", "bbg_multi_ctx", "bbg_multi_sync", "bbg_multi_srs_register",
    "bbg_multi_srs_synth_hashed", "bbg_multi_srs_num_points", "bbg_multi_msm", "bbg_multi_ntt_device", "bbg_multi_ntt", "bbg_multi_set_option",
]


class MemoryInfo(ctypes.Structure):
    """bbg_memory_info (include/bbg.h)."""
    _fields_ = [(k, ctypes.c_size_t) for k in ("srs_points", "srs_tables", "ntt_tables", "msm_arena", "scratch", "prover_keys", "total",
                                               "device_total", "device_free")] + \
               [(k, ctypes.c_uint) for k in ("live_srs", "live_provers", "ntt_domains")]


def _u64(a, shape_last):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim == 1:
        a = a.reshape(-1, shape_last)
    if a.shape[-1] != shape_last:
        raise ValueError(f"expected (*, {shape_last}) uint64 limbs, got {a.shape}")
    return a


class Srs:
    def __init__(self, owner, handle):
        self._owner, self.handle = owner, handle

    @property
    def num_points(self):
        return int(self._owner.lib.bbg_srs_num_points(self.handle))

    def read(self, start=0, count=None):
        count = self.num_points - start if count is None else count
        out = np.empty((count, 8), dtype=np.uint64)
        self._owner._ck(self._owner.lib.bbg_srs_read(self.handle, start, count, out.ctypes.data))
        return out

    def write_transcript(self, directory, points_per_file=0, g2_x_raw=None):
        """Ignition-format files directory/transcriptNN.dat holding points 1 .. n-1 (bbg_srs_write_transcript)."""
        g2 = None if g2_x_raw is None else ctypes.create_string_buffer(bytes(g2_x_raw), 128)
        self._owner._ck(self._owner.lib.bbg_srs_write_transcript(self.handle, str(directory).encode(), points_per_file, g2))

    def free(self):
        if self.handle:
            self._owner.lib.bbg_srs_free(self.handle)
            self.handle = None


class Bbg:
    """One context per GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device=0):
        self.lib = load_library()
        if self.lib.bbg_device_count() <= 0:
            raise BbgError("no HIP device visible: libbbg has no CPU fallback")
        h = ctypes.c_void_p()
        rc = self.lib.bbg_init(device, ctypes.byref(h))
        if rc != 0:
            raise BbgError(f"bbg_init failed ({rc}): {self.lib.bbg_last_error().decode()}")
        self.ctx = h
        self.device = device

    def _ck(self, rc):
        if rc != 0:
            raise BbgError(f"libbbg error {rc}: {self.lib.bbg_last_error().decode()}")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.bbg_destroy(self.ctx)
            self.ctx = None

    def sync(self):
        self._ck(self.lib.bbg_sync(self.ctx))

    def join(self, lag=0):
        self._ck(self.lib.bbg_join_lag(self.ctx, lag) if lag else self.lib.bbg_join(self.ctx))

    def set_stream(self, stream_ptr):
        self._ck(self.lib.bbg_set_stream(self.ctx, ctypes.c_void_p(stream_ptr)))

    def set_option(self, key, value):
        self._ck(self.lib.bbg_set_option(self.ctx, key.encode(), int(value)))

    # ---- SRS
    def srs_register(self, points, stride_bytes=64):
        pts = np.ascontiguousarray(points, dtype=np.uint64)
        n = pts.size * 8 // stride_bytes
        h = ctypes.c_void_p()
        self._ck(self.lib.bbg_srs_register(self.ctx, pts.ctypes.data, n, stride_bytes, ctypes.byref(h)))
        return Srs(self, h)

    def srs_register_device(self, d_points, n):
        h = ctypes.c_void_p()
        self._ck(self.lib.bbg_srs_register_device(self.ctx, ctypes.c_void_p(d_points), n, ctypes.byref(h)))
        return Srs(self, h)

    def srs_synth_linear(self, a, s, n):
        h = ctypes.c_void_p()
        self._ck(self.lib.bbg_srs_synth_linear(self.ctx, a, s, n, ctypes.byref(h)))
        return Srs(self, h)

    def srs_synth_hashed(self, seed, n):
        h = ctypes.c_void_p()
        self._ck(self.lib.bbg_srs_synth_hashed(self.ctx, seed, n, ctypes.byref(h)))
        return Srs(self, h)

    def srs_load_transcript(self, directory, num_points):
        h = ctypes.c_void_p()
        self._ck(self.lib.bbg_srs_load_transcript(self.ctx, str(directory).encode(), num_points, ctypes.byref(h)))
        return Srs(self, h)

    # ---- MSM
    def msm(self, srs, scalars, start=0):
        sc = _u64(scalars, 4)
        out = np.zeros(12, dtype=np.uint64)
        self._ck(self.lib.bbg_msm(self.ctx, srs.handle, sc.ctypes.data, start, sc.shape[0], out.ctypes.data))
        return out

    def msm_device(self, srs, d_scalars, n, d_out, start=0):
        self._ck(self.lib.bbg_msm_device(self.ctx, srs.handle, ctypes.c_void_p(d_scalars), start, n, ctypes.c_void_p(d_out)))

    def msm_batch(self, srs, scalar_list, starts=None):
        """bbg_msm_batch: len(scalar_list) MSMs over one SRS through ONE launch set; returns (count, 12) Jacobians."""
        scs = [_u64(s, 4) for s in scalar_list]
        cnt = len(scs)
        ptrs = (ctypes.c_void_p * cnt)(*[ctypes.c_void_p(s.ctypes.data) for s in scs])
        ns = (ctypes.c_size_t * cnt)(*[s.shape[0] for s in scs])
        fr = None if starts is None else (ctypes.c_size_t * cnt)(*[int(v) for v in starts])
        out = np.zeros((cnt, 12), dtype=np.uint64)
        self._ck(self.lib.bbg_msm_batch(self.ctx, srs.handle, cnt, ptrs, fr, ns, out.ctypes.data))
        return out

    def msm_batch_device(self, srs, d_scalar_ptrs, ns, d_out, starts=None):
        cnt = len(d_scalar_ptrs)
        ptrs = (ctypes.c_void_p * cnt)(*[ctypes.c_void_p(int(p)) for p in d_scalar_ptrs])
        nn = (ctypes.c_size_t * cnt)(*[int(v) for v in ns])
        fr = None if starts is None else (ctypes.c_size_t * cnt)(*[int(v) for v in starts])
        self._ck(self.lib.bbg_msm_batch_device(self.ctx, srs.handle, cnt, ptrs, fr, nn, ctypes.c_void_p(d_out)))

    def ntt_plan(self, log2n):
        """{passes, log_radix, kernel, tile_log} of a 2^log2n transform as it would run now (bbg_ntt_plan)."""
        cint = ctypes.c_int
        passes, kernel, tile = cint(), cint(), cint()
        radix = (cint * 4)()
        self._ck(self.lib.bbg_ntt_plan(self.ctx, log2n, ctypes.byref(passes), radix, ctypes.byref(kernel), ctypes.byref(tile)))
        names = {0: "k_ntt_pass", 8: "k_ntt_pass8", 81: "k_ntt_pass8s", 29: "k_ntt_pass29"}
        return {"passes": passes.value, "log_radix": [radix[q] for q in range(passes.value)], "kernel": names.get(kernel.value, str(kernel.value)),
                "tile_log": tile.value}

    def msm_plan(self, n, srs=None):
        """(window bits C, windows) an n-term MSM would run with now (bbg_msm_plan)."""
        c, w = ctypes.c_int(), ctypes.c_int()
        self._ck(self.lib.bbg_msm_plan(self.ctx, srs.handle if srs is not None else None, n, ctypes.byref(c), ctypes.byref(w)))
        return c.value, w.value

    def memory_report(self):
        """bbg_memory_report as a dict (bytes by purpose, live handle counts, hipMemGetInfo)."""
        info = MemoryInfo()
        self._ck(self.lib.bbg_memory_report(self.ctx, ctypes.byref(info)))
        return {k: int(getattr(info, k)) for k, _ in MemoryInfo._fields_}

    def memory_trim(self, tables=False):
        rel = ctypes.c_size_t()
        self._ck(self.lib.bbg_memory_trim(self.ctx, 1 if tables else 0, ctypes.byref(rel)))
        return int(rel.value)

    def g1_sum(self, jacobians):
        j = _u64(jacobians, 12)
        out = np.zeros(12, dtype=np.uint64)
        self._ck(self.lib.bbg_g1_sum(self.ctx, j.ctypes.data, j.shape[0], out.ctypes.data))
        return out

    def g1_sum_device(self, d_jacobians, n, d_out):
        self._ck(self.lib.bbg_g1_sum_device(self.ctx, ctypes.c_void_p(d_jacobians), n, ctypes.c_void_p(d_out)))

    def g1_normalize(self, jacobians):
        j = _u64(jacobians, 12)
        out = np.zeros((j.shape[0], 8), dtype=np.uint64)
        self._ck(self.lib.bbg_g1_normalize(self.ctx, j.ctypes.data, j.shape[0], out.ctypes.data))
        return out

    # ---- NTT
    def ntt(self, coeffs, op=FFT, generator_size=0, constant=None):
        a = _u64(coeffs, 4).copy()
        n = a.shape[0]
        log2n = n.bit_length() - 1
        if n == 0 or (1 << log2n) != n:
            raise ValueError("NTT size must be a power of two")
        c = None if constant is None else np.ascontiguousarray(constant, dtype=np.uint64)
        self._ck(self.lib.bbg_ntt(self.ctx, a.ctypes.data, log2n, op, generator_size, None if c is None else c.ctypes.data))
        return a

    def coset_fft_extend(self, coeffs, log2_domain):
        """The prover's FFT work item (work_queue.hpp:252-264): n coefficients -> 2^log2_domain + 4 coset evaluations."""
        a = _u64(coeffs, 4)
        n = a.shape[0]
        log2n = n.bit_length() - 1
        if n == 0 or (1 << log2n) != n:
            raise ValueError("coefficient count must be a power of two")
        out = np.empty(((1 << log2_domain) + 4, 4), dtype=np.uint64)
        self._ck(self.lib.bbg_coset_fft_extend(self.ctx, a.ctypes.data, log2n, log2_domain, out.ctypes.data))
        return out

    def poly_linear_combination_device(self, d_polys, scalars, d_base, d_out, n):
        """out = base + sum_k polys[k] * scalars[k] (d_base may be 0/None)."""
        arr = (ctypes.c_void_p * max(len(d_polys), 1))(*[ctypes.c_void_p(int(p)) for p in d_polys])
        sc = _u64(scalars, 4) if len(d_polys) else np.zeros((1, 4), dtype=np.uint64)
        self._ck(self.lib.bbg_poly_linear_combination_device(self.ctx, arr, sc.ctypes.data, len(d_polys),
                                                             ctypes.c_void_p(d_base) if d_base else None, ctypes.c_void_p(d_out), n))

    def permutation_grand_product_device(self, d_wires, d_sigmas, log2n, beta, gamma, ks, d_z):
        w = (ctypes.c_void_p * 4)(*[ctypes.c_void_p(int(p)) for p in d_wires])
        s_ = (ctypes.c_void_p * 4)(*[ctypes.c_void_p(int(p)) for p in d_sigmas])
        ch = np.ascontiguousarray(np.concatenate([np.reshape(beta, (1, 4)), np.reshape(gamma, (1, 4)), np.reshape(ks, (3, 4))]), dtype=np.uint64)
        self._ck(self.lib.bbg_permutation_grand_product_device(self.ctx, w, s_, log2n, ch.ctypes.data, ctypes.c_void_p(d_z)))

    def quotient_widget_device(self, widget, d_polys, log2_large, challenges, d_quotient):
        """d_polys: list of BBG_QP_COUNT device addresses (0 = not supplied), BBG_QP_EXT_COUNT = 23 for the MiMC widget (7);
        challenges: (9, 4) uint64.  Returns the next alpha_base."""
        padded = list(d_polys) + [0] * max(0, 23 - len(d_polys))  # the library may read the extended table: never hand it a short array
        arr = (ctypes.c_void_p * len(padded))(*[ctypes.c_void_p(int(p)) if p else None for p in padded])
        ch = np.ascontiguousarray(challenges, dtype=np.uint64)
        assert ch.shape == (9, 4)
        out = np.zeros(4, dtype=np.uint64)
        self._ck(self.lib.bbg_quotient_widget_device(self.ctx, widget, arr, log2_large, ch.ctypes.data, ctypes.c_void_p(d_quotient),
                                                     out.ctypes.data))
        return out

    def ntt_device(self, d_coeffs, log2n, op=FFT, generator_size=0, constant=None):
        c = None if constant is None else np.ascontiguousarray(constant, dtype=np.uint64)
        self._ck(self.lib.bbg_ntt_device(self.ctx, ctypes.c_void_p(d_coeffs), log2n, op, generator_size,
                                         None if c is None else c.ctypes.data))

    def ntt_prepare(self, log2n):
        self._ck(self.lib.bbg_ntt_prepare(self.ctx, log2n))

    def coset_fft_split(self, coeffs, ext):
        a = _u64(coeffs, 4)
        n = a.shape[0]
        log2n = n.bit_length() - 1
        buf = np.zeros((n * ext, 4), dtype=np.uint64)
        buf[:n] = a
        self._ck(self.lib.bbg_coset_fft_split(self.ctx, buf.ctypes.data, log2n, ext))
        return buf

    # ---- sharded-NTT building blocks
    def scale_powers_device(self, d_a, count, base, start=None):
        b = np.ascontiguousarray(base, dtype=np.uint64)
        s_ = None if start is None else np.ascontiguousarray(start, dtype=np.uint64)
        self._ck(self.lib.bbg_scale_powers_device(self.ctx, ctypes.c_void_p(d_a), count, None if s_ is None else s_.ctypes.data,
                                                  b.ctypes.data))

    def fr_root_pow(self, log2n, e, inverse=False):
        out = np.zeros(4, dtype=np.uint64)
        self._ck(self.lib.bbg_fr_root_pow(self.ctx, log2n, e, 1 if inverse else 0, out.ctypes.data))
        return out

    def fr_pow(self, base, e):
        b = np.ascontiguousarray(base, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        self._ck(self.lib.bbg_fr_pow(self.ctx, b.ctypes.data, e, out.ctypes.data))
        return out

    def cross_dft_device(self, d_in, d_out, log2g, length, log2n, inverse=False):
        self._ck(self.lib.bbg_cross_dft_device(self.ctx, ctypes.c_void_p(d_in), ctypes.c_void_p(d_out), log2g, length, log2n,
                                               1 if inverse else 0))

    # ---- polynomial helpers (device pointers)
    def poly_op_device(self, op, d_a, d_b, d_r, n):
        self._ck(self.lib.bbg_poly_op_device(self.ctx, op, ctypes.c_void_p(d_a), ctypes.c_void_p(d_b), ctypes.c_void_p(d_r), n))

    def poly_evaluate_device(self, d_coeffs, n, z):
        zz = np.ascontiguousarray(z, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        self._ck(self.lib.bbg_poly_evaluate_device(self.ctx, ctypes.c_void_p(d_coeffs), n, zz.ctypes.data, out.ctypes.data))
        return out

    def kate_opening_device(self, d_src, d_dest, n, z):
        zz = np.ascontiguousarray(z, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        self._ck(self.lib.bbg_kate_opening_device(self.ctx, ctypes.c_void_p(d_src), ctypes.c_void_p(d_dest), n, zz.ctypes.data,
                                                  out.ctypes.data))
        return out

    # host-buffer forms (what the C++ shim binds evaluate / compute_kate_opening_coefficients / divide_by_pseudo_vanishing_polynomial to)
    def poly_evaluate(self, coeffs, z):
        a = _u64(coeffs, 4)
        zz = np.ascontiguousarray(z, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        self._ck(self.lib.bbg_poly_evaluate(self.ctx, a.ctypes.data, a.shape[0], zz.ctypes.data, out.ctypes.data))
        return out

    def kate_opening(self, src, z, in_place=False):
        """Returns (dest, F(z)); in_place=True passes dest == src the way KateCommitmentScheme::batch_open does."""
        a = _u64(src, 4).copy()
        d = a if in_place else np.empty_like(a)
        zz = np.ascontiguousarray(z, dtype=np.uint64)
        f = np.zeros(4, dtype=np.uint64)
        self._ck(self.lib.bbg_kate_opening(self.ctx, a.ctypes.data, d.ctypes.data, a.shape[0], zz.ctypes.data, f.ctypes.data))
        return d, f

    def divide_by_pseudo_vanishing(self, evals, log2_src, num_roots_cut=4):
        a = _u64(evals, 4).copy()
        log2_target = a.shape[0].bit_length() - 1
        self._ck(self.lib.bbg_divide_by_pseudo_vanishing(self.ctx, a.ctypes.data, log2_src, log2_target, num_roots_cut))
        return a

    def divide_by_pseudo_vanishing_device(self, d_evals, log2_src, log2_target, num_roots_cut=4):
        self._ck(self.lib.bbg_divide_by_pseudo_vanishing_device(self.ctx, ctypes.c_void_p(d_evals), log2_src, log2_target, num_roots_cut))

    # ---- raw device memory (for hosts without torch)
    def dev_alloc(self, nbytes):
        p = ctypes.c_void_p()
        self._ck(self.lib.bbg_dev_alloc(self.ctx, nbytes, ctypes.byref(p)))
        return p.value

    def dev_free(self, ptr):
        self._ck(self.lib.bbg_dev_free(self.ctx, ctypes.c_void_p(ptr)))

    def dev_upload(self, ptr, array):
        a = np.ascontiguousarray(array)
        self._ck(self.lib.bbg_dev_upload(self.ctx, ctypes.c_void_p(ptr), a.ctypes.data, a.nbytes))

    def dev_download(self, ptr, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        self._ck(self.lib.bbg_dev_download(self.ctx, out.ctypes.data, ctypes.c_void_p(ptr), out.nbytes))
        return out

    def profile_enable(self, on=True):
        self._ck(self.lib.bbg_profile_enable(self.ctx, 1 if on else 0))

    def profile_get(self, name):
        """(total_ms, launches) of the named kernel since profile_enable(True)."""
        ms, cnt = ctypes.c_double(0), ctypes.c_size_t(0)
        self._ck(self.lib.bbg_profile_get(self.ctx, name.encode(), ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    def field_op(self, which, op, a, b=None):
        a = _u64(a, 4)
        out = np.empty_like(a)
        bp = None
        if b is not None:
            b = _u64(b, 4)
            bp = b.ctypes.data
        self._ck(self.lib.bbg_field_op(self.ctx, which, op, a.ctypes.data, bp, out.ctypes.data, a.shape[0]))
        return out
