"""TEST TOOLING (not part of the product): Python stand-ins for work-queue callbacks, used to put single pieces of libbbg.so behind the
reference prover's rounds and compare them with the reference item by item (tests/test_gpu_parity.py, tests/tools/bench_prover_real.py).
The product's resident prover is C++: shim/bbg_resident_prover.hpp over include/bbg.h's bbg_prover_* (no Python, no torch).

An engine instance belongs to ONE RefProver session (one proving key): its device copies of per-key arrays are keyed by the host
address inside that session only and die with the engine -- there is no content fingerprinting.

Works on raw host addresses of the prover's own buffers, in place:

    msm_raw(scalars, count, out)                    work_queue SCALAR_MULTIPLICATION (work_queue.hpp:218-245) -> bbg_msm
    fft_item_raw(wire, log2n, wire_fft, log2_4n)    work_queue FFT (:252-264)                                 -> bbg_coset_fft_extend
    coset_fft_raw / ifft_raw                                                                                  -> bbg_ntt
    round3_raw(wires, sigmas, challenges, blind, log2n, z)   the permutation polynomial of execute_third_round
                                                    (permutation_widget_impl.hpp:48-297): grand product, blinded rows, ifft
    round4_raw(poly_ptrs, challenges, log2n, q)     the quotient of execute_fourth_round (prover.cpp:304-343): the five TurboPLONK
                                                    widgets, divide_by_pseudo_vanishing_polynomial and coset_ifft on the device;
                                                    selector / sigma / L_1 arrays are uploaded once per proving key

"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as _ge  # noqa: E402

binding = _ge.load_package().binding


class WorkQueueEngine:
    """MSM / coset FFT / iFFT items through the host entry points (the FFT item as copy + coset_fft + 4 appended values on the
    prover's side, exactly as work_queue::process_queue does it)."""
    raw = True

    def __init__(self, bbg, srs):
        self.bbg, self.srs = bbg, srs

    def msm_raw(self, scalars, count, out):
        b = self.bbg
        b._ck(b.lib.bbg_msm(b.ctx, self.srs.handle, ctypes.c_void_p(scalars), 0, count, ctypes.c_void_p(out)))

    def coset_fft_raw(self, coeffs, log2_domain, generator_size):
        b = self.bbg
        b._ck(b.lib.bbg_ntt(b.ctx, ctypes.c_void_p(coeffs), log2_domain, binding.COSET_FFT, generator_size, None))

    def ifft_raw(self, coeffs, log2n):
        b = self.bbg
        b._ck(b.lib.bbg_ntt(b.ctx, ctypes.c_void_p(coeffs), log2n, binding.IFFT, 0, None))


class FusedFftEngine(WorkQueueEngine):
    """+ the whole FFT work item as one call (n coefficients up, 4n + 4 values down)."""

    def fft_item_raw(self, wire, log2n, wire_fft, log2_domain):
        b = self.bbg
        b._ck(b.lib.bbg_coset_fft_extend(b.ctx, ctypes.c_void_p(wire), log2n, log2_domain, ctypes.c_void_p(wire_fft)))


class Round4Engine(FusedFftEngine):
    """+ the quotient of execute_fourth_round on the device.  With queue_via_reference = True only round 4 is taken over and
    the work queue is left to the prover's own process_queue (for a prover that is already linked against the shim)."""
    queue_via_reference = False

    def __init__(self, bbg, srs):
        super().__init__(bbg, srs)
        self._static = {}  # per proving key: device copies of the arrays that do not change between proofs

    def _upload(self, host_ptr, count):
        import torch
        t = torch.empty(count * 4, dtype=torch.int64, device="cuda")
        b = self.bbg
        b._ck(b.lib.bbg_dev_upload(b.ctx, ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(host_ptr), count * 32))
        return t

    def round4_raw(self, poly_ptrs, challenges, log2n, quotient_ptr):
        import torch
        b = self.bbg
        m = 4 << log2n
        # sigma_1..4, q_*, L_1 on the coset are fixed per proving key: uploaded once per engine (= per session, see the module header)
        key = (log2n,) + tuple(poly_ptrs[5:])
        if key not in self._static:
            self._static = {key: [self._upload(p, m) for p in poly_ptrs[5:]]}
        wires = [self._upload(p, m) for p in poly_ptrs[:5]]  # w_1..4, z: new for every proof
        dev = wires + self._static[key]
        ptrs = [t.data_ptr() for t in dev]
        quot = torch.empty(m * 4, dtype=torch.int64, device="cuda")
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).copy()
        alpha_base = ch[0].copy()
        for widget in range(5):  # the prover's order: permutation, arithmetic, fixed base, range, logic
            ch[0] = alpha_base
            alpha_base = b.quotient_widget_device(widget, ptrs, log2n + 2, ch, quot.data_ptr())
        b.divide_by_pseudo_vanishing_device(quot.data_ptr(), log2n, log2n + 2, 4)
        b.ntt_device(quot.data_ptr(), log2n + 2, binding.COSET_IFFT)
        b._ck(b.lib.bbg_dev_download(b.ctx, ctypes.c_void_p(quotient_ptr), ctypes.c_void_p(quot.data_ptr()), m * 32))


class Round34Engine(Round4Engine):
    """+ the permutation polynomial z of execute_third_round on the device (sigma permutations resident per proving key)."""

    def __init__(self, bbg, srs):
        super().__init__(bbg, srs)
        self._sigma = {}

    def round3_raw(self, wire_ptrs, sigma_ptrs, challenges, blind, log2n, z_ptr):
        import torch
        b = self.bbg
        n = 1 << log2n
        key = (log2n,) + tuple(sigma_ptrs)
        if key not in self._sigma:
            self._sigma = {key: [self._upload(p, n) for p in sigma_ptrs]}
        wires = [self._upload(p, n) for p in wire_ptrs]
        z = torch.empty(n * 4, dtype=torch.int64, device="cuda")
        ch = np.ascontiguousarray(challenges, dtype=np.uint64)
        b.permutation_grand_product_device([t.data_ptr() for t in wires], [t.data_ptr() for t in self._sigma[key]], log2n, ch[0], ch[1],
                                           ch[2:5], z.data_ptr())
        # rows n-3 .. n-1 carry the prover's zero-knowledge blinding (4 roots are cut out of the vanishing polynomial)
        bl = np.ascontiguousarray(blind, dtype=np.uint64)
        b._ck(b.lib.bbg_dev_upload(b.ctx, ctypes.c_void_p(z.data_ptr() + (n - 3) * 32), bl.ctypes.data, 96))
        b.ntt_device(z.data_ptr(), log2n, binding.IFFT)
        b._ck(b.lib.bbg_dev_download(b.ctx, ctypes.c_void_p(z_ptr), ctypes.c_void_p(z.data_ptr()), n * 32))


class Round346Engine(Round34Engine):
    """+ the two opening polynomials of execute_sixth_round / KateCommitmentScheme::batch_open (kate_commitment_scheme.cpp:133-236):
    F = t_low + sum nu_k P_k (bbg_poly_linear_combination_device), W = (F - F(z)) / (X - z) (bbg_kate_opening_device).  Selector and
    permutation polynomials (coefficient form) are resident per proving key; wires, z, the quotient parts and r(X) go up per proof."""

    def __init__(self, bbg, srs):
        super().__init__(bbg, srs)
        self._coeff = {}

    def _resident(self, host_ptr, n):
        """Device copy of a coefficient-form polynomial of THIS session; `fresh` addresses (per-proof arrays) are never cached."""
        fp = (host_ptr, n)
        t = self._coeff.get(fp)
        if t is None:
            if len(self._coeff) > 64:
                self._coeff.clear()
            t = self._coeff[fp] = Round4Engine._upload(self, host_ptr, n)
        return t

    def round6_raw(self, polys_zeta, scalars_zeta, base, polys_omega, scalars_omega, zeta, zeta_omega, n, w_zeta, w_zeta_omega):
        import torch
        b = self.bbg
        dz = [self._resident(p, n) for p in polys_zeta]
        do = [self._resident(p, n) for p in polys_omega]
        dbase = Round4Engine._upload(self, base, n)
        f = torch.empty(n * 4, dtype=torch.int64, device="cuda")
        w = torch.empty(n * 4, dtype=torch.int64, device="cuda")
        b.poly_linear_combination_device([t.data_ptr() for t in dz], scalars_zeta, dbase.data_ptr(), f.data_ptr(), n)
        b.kate_opening_device(f.data_ptr(), w.data_ptr(), n, zeta)
        b._ck(b.lib.bbg_dev_download(b.ctx, ctypes.c_void_p(w_zeta), ctypes.c_void_p(w.data_ptr()), n * 32))
        b.poly_linear_combination_device([t.data_ptr() for t in do], scalars_omega, None, f.data_ptr(), n)
        b.kate_opening_device(f.data_ptr(), w.data_ptr(), n, zeta_omega)
        b._ck(b.lib.bbg_dev_download(b.ctx, ctypes.c_void_p(w_zeta_omega), ctypes.c_void_p(w.data_ptr()), n * 32))


class ResidentEngine(Round346Engine):
    """+ the coset FFTs of the work queue stay ON THE DEVICE: with rounds 3 and 4 computed there, nothing on the host reads the
    4n-point "*_fft" arrays any more, so the FFT work item neither downloads its 128 MiB result nor is it uploaded again for the
    quotient.  (The host arrays are left untouched: only valid together with round4_raw.)"""

    def __init__(self, bbg, srs):
        import torch
        super().__init__(bbg, srs)
        self._fft = {}  # host address of a wire_fft array -> device tensor holding its 4n + 4 values
        # this engine mixes torch kernels (zero fill, slice copy) with library kernels on the same buffers: one stream for both
        bbg.set_stream(torch.cuda.current_stream().cuda_stream)

    def fft_item_raw(self, wire, log2n, wire_fft, log2_domain):
        import torch
        b = self.bbg
        n, m = 1 << log2n, 1 << log2_domain
        t = torch.zeros((m + 4) * 4, dtype=torch.int64, device="cuda")
        b._ck(b.lib.bbg_dev_upload(b.ctx, ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(wire), n * 32))
        b.ntt_device(t.data_ptr(), log2_domain, binding.COSET_FFT, n)
        t[m * 4:] = t[:16]  # add_lagrange_base_coefficient x4: the first four values again at 4n .. 4n+3
        self._fft[wire_fft] = t

    def _upload(self, host_ptr, count):
        if host_ptr in self._fft and self._fft[host_ptr].numel() >= count * 4:
            return self._fft.pop(host_ptr)  # produced on the device by fft_item_raw: no transfer
        return super()._upload(host_ptr, count)
