"""Multi-GPU decomposition of the hot path: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI
on the GPU box, "gloo" in the CPU tests).

MSM shards by POINT RANGE -- the reference's own precedent is Pippenger::pippenger_unsafe(scalars, from, range) +
g1_sum (ecc/curves/bn254/scalar_multiplication/pippenger.cpp:27-31, c_bind.cpp:31-46): rank g owns SRS points and
scalars [g*n/G, (g+1)*n/G), runs the whole bucket MSM locally, and the G 96-byte Jacobian partials are all-gathered
and added.  RCCL has no elliptic-curve reduction operator, so "reduce" = all_gather + local g1_sum; the exchange is
G*96 bytes, i.e. latency only.  The NTT side of the prover needs no exchange at all: a 4n coset FFT of n non-zero
coefficients is 4 independent size-n coset FFTs (work_queue.hpp:166-199), so ranks take whole transforms.
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous point range of `rank`: (start, count); the last rank takes the remainder."""
    base = n // world
    start = rank * base
    count = base if rank < world - 1 else n - start
    return start, count


def all_gather_partials(local_jacobian, dist, device=None):
    """local_jacobian: 12 uint64 limbs (numpy) or a torch int64 tensor of 12.  Returns a (world, 12) uint64 numpy array
    holding every rank's partial, identical on all ranks."""
    import torch
    if isinstance(local_jacobian, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(local_jacobian, dtype=np.uint64).view(np.int64).copy())
        if device is not None:
            t = t.to(device)
    else:
        t = local_jacobian
    world = dist.get_world_size()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy().view(np.uint64)


def msm_sharded(bbg, srs_local, d_scalars_ptr, n_local, d_out_tensor, dist):
    """Local bucket MSM on this rank's shard, then all_gather + g1_sum.  d_out_tensor: torch int64[12] on the GPU.
    Returns the global result as 12 uint64 limbs (identical on every rank)."""
    bbg.msm_device(srs_local, d_scalars_ptr, n_local, d_out_tensor.data_ptr())
    bbg.sync()
    if dist is None or dist.get_world_size() == 1:
        return d_out_tensor.cpu().numpy().view(np.uint64)
    parts = all_gather_partials(d_out_tensor, dist)
    return bbg.g1_sum(parts)


class BbgOps:
    """Adapter: the three device operations the sharded pipeline needs, on torch CUDA tensors.  `bbg_side`: a second context whose stream is
    the pipeline's side stream (the group sum of step i-1 then runs beside step i instead of in front of its NTT and the next sort)."""

    def __init__(self, bbg, srs, bbg_side=None):
        self.bbg, self.srs, self.bbg_side = bbg, srs, bbg_side

    def msm(self, d_scalars, n, d_out):
        self.bbg.msm_device(self.srs, d_scalars.data_ptr(), n, d_out.data_ptr())

    def join(self, lag):
        self.bbg.join(lag)

    def g1_sum(self, d_jacobians, count, d_out):
        (self.bbg_side or self.bbg).g1_sum_device(d_jacobians.data_ptr(), count, d_out.data_ptr())


class ShardedMsmPipeline:
    """One global MSM per submit(), sharded by point range over the ranks of `dist`, software-pipelined: while rank-local MSM i is still in its
    bucket-reduction phase (auxiliary stream), the 96-byte partials of earlier MSMs are all-gathered (RCCL) and summed.  flush() completes what
    is outstanding.  Everything is stream-ordered; nothing returns to the host.  results[i % (2 * depth)] holds global result i after its
    finish step; last_result() is the newest.

    `depth`: partials are exchanged `depth` MSMs at a time (ONE all-gather of depth x 96 bytes and `depth` group sums): the exchange is a handful of
    microsecond-sized operations whose fixed cost -- not their bytes -- is what a step pays (0.08 ms of a 1.48 ms step with a world of one).
    `side_stream` (a torch.cuda.Stream, GPU runs): the exchange is issued there, ordered behind the local MSMs by an event, so that the main
    stream goes straight on with whatever the caller queues next (the bench step's NTT and the next MSM's sort)."""

    def __init__(self, ops, dist, new_tensor, side_stream=None, depth=1, cuda=None):
        """`cuda`: the stream / event API (default torch.cuda); the CPU tests pass a recording stand-in to check the ordering protocol."""
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if side_stream is not None and getattr(ops, "bbg_side", self) is None:
            # the group sums would run on the MAIN context's stream while the all-gather that feeds them runs on the side stream: a race
            raise ValueError("side_stream needs ops with a side context bound to it (BbgOps(bbg, srs, bbg_side))")
        self.ops, self.dist, self.depth = ops, dist, depth
        self.world = dist.get_world_size() if dist is not None else 1
        self.ring = 2 * depth
        self.partial_block = new_tensor(12 * self.ring)
        self.partial = [self.partial_block[12 * k:12 * k + 12] for k in range(self.ring)]
        self.gathered = new_tensor(12 * depth * self.world)
        self.results = [new_tensor(12) for _ in range(self.ring)]
        self.side = side_stream
        if side_stream is not None:
            if cuda is None:
                import torch
                cuda = torch.cuda
            self._cuda = cuda
            self.ev_ready = [cuda.Event(), cuda.Event()]  # main stream: the partials of this half of the ring are final
            self.ev_done = [cuda.Event(), cuda.Event()]   # side stream: this half's partials read, its results written
        self.reset()

    def reset(self):
        """Forget finished work (call only after flush() and a device synchronisation)."""
        self.count = 0     # ring position of the next MSM (= MSMs submitted, plus the slots flush() skipped to end a partial batch)
        self.finished = 0  # ring position up to which global results have been issued
        self.submitted = 0  # MSMs submitted since reset()
        self.last_slot = None
        self.done_valid = [False, False]

    def last_result(self):
        return self.results[self.last_slot] if self.last_slot is not None else None

    def _combine(self, j0, cnt):
        lo = 12 * (j0 % self.ring)
        block = self.partial_block[lo:lo + 12 * cnt]
        if self.dist is not None:  # also with a world of one (BBG_FORCE_DIST=1): the emulation makes every call the N > 1 path makes
            gathered = self.gathered[:12 * cnt * self.world]
            self.dist.all_gather_into_tensor(gathered, block)
            if cnt == 1:
                self.ops.g1_sum(gathered, self.world, self.results[j0 % self.ring])
            else:  # [rank][k][12] -> [k][rank][12]: each MSM's partials contiguous for the group sum
                per_msm = gathered.view(self.world, cnt, 12).transpose(0, 1).contiguous()
                for k in range(cnt):
                    self.ops.g1_sum(per_msm[k].reshape(-1), self.world, self.results[(j0 + k) % self.ring])
        else:
            for k in range(cnt):
                self.results[(j0 + k) % self.ring].copy_(self.partial[(j0 + k) % self.ring])

    def _finish(self, j0, cnt, lag):
        """Global results of MSMs j0 .. j0 + cnt - 1 (one half of the ring, or the head of one)."""
        assert j0 % self.depth == 0 and cnt <= self.depth, "a batch is one half of the ring"
        self.ops.join(lag)  # device-side wait for the reductions of everything but the `lag` most recent MSMs
        if self.side is None:
            self._combine(j0, cnt)
            return
        cuda, h = self._cuda, (j0 // self.depth) & 1
        self.ev_ready[h].record(cuda.current_stream())
        self.side.wait_event(self.ev_ready[h])
        with cuda.stream(self.side):
            self._combine(j0, cnt)
            self.ev_done[h].record(self.side)
        self.done_valid[h] = True

    def submit(self, d_scalars, n):
        """Issues the local MSM; returns the index into `results` where this MSM's global result appears (valid after the batch it belongs
        to has been finished -- at the latest after flush() -- until 2 * depth further submits)."""
        i = self.count
        if self.side is not None and i % self.depth == 0:  # MSM i starts overwriting a half of the ring: the side stream must be done with it
            h = (i // self.depth) & 1
            if self.done_valid[h]:
                self._cuda.current_stream().wait_event(self.ev_done[h])
        self.ops.msm(d_scalars, n, self.partial[i % self.ring])
        self.count += 1
        self.submitted += 1
        self.last_slot = i % self.ring
        if self.count - self.finished == self.depth + 1:  # a whole batch lies behind the MSM just issued
            self._finish(self.finished, self.depth, 1)
            self.finished += self.depth
        return self.last_slot

    def flush(self):
        lag = 0
        while self.finished < self.count:
            cnt = min(self.depth, self.count - self.finished)
            self._finish(self.finished, cnt, lag)  # the first call waits for every outstanding reduction
            self.finished += cnt
        # A flushed remainder ends its batch: the next submit starts a fresh half of the ring.  (Continuing inside the partial batch would
        # make the following batch straddle the ring's end -- a truncated all-gather block -- and break the half <-> event pairing.)
        self.count = self.finished = -(-self.count // self.depth) * self.depth
        if self.side is not None and self.count:
            self._cuda.current_stream().wait_stream(self.side)
        return self.last_result()


class HostStagedDist:
    """The subset of torch.distributed the sharded paths use, with every payload staged through host memory: the REHEARSAL carrier of
    `bench.py --gpus N` when all N ranks share ONE device (BBG_DIST_ONE_DEVICE=1; RCCL refuses a communicator with duplicate devices, gloo
    has no device-side all-to-all).  A collective here = wait for the current stream, copy the operand to the host, run the gloo collective,
    copy the result back on the current stream -- the same program order, the same shapes and the same arithmetic as the RCCL run, only the
    wire is different (and blocking: timings taken this way say nothing about xGMI).  With CPU tensors the staging copies are no-ops, which
    is how tests/test_distributed_cpu.py drives it."""

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def get_world_size(self): return self._d.get_world_size()

    def get_rank(self): return self._d.get_rank()

    def barrier(self): self._d.barrier()

    def destroy_process_group(self): self._d.destroy_process_group()

    @staticmethod
    def _host(t):
        return t.detach().cpu().contiguous()  # .cpu() orders itself behind the work queued on the current stream

    def all_reduce(self, t, op=None):
        h = self._host(t)
        self._d.all_reduce(h, op=op if op is not None else self._d.ReduceOp.SUM)
        t.copy_(h)

    def all_gather_into_tensor(self, out, t):
        h = self._host(t).reshape(-1)
        parts = [h.new_empty(h.shape) for _ in range(self.get_world_size())]
        self._d.all_gather(parts, h)
        import torch
        out.copy_(torch.cat(parts).reshape(out.shape))

    def all_gather(self, outs, t):
        h = self._host(t)
        parts = [h.new_empty(h.shape) for _ in outs]
        self._d.all_gather(parts, h)
        for o, p in zip(outs, parts):
            o.copy_(p)

    def all_to_all_single(self, out, t):
        """Chunk s of `t` goes to rank s; chunk s of `out` comes from rank s (equal splits).  gloo's all-to-all exists for CPU tensors."""
        h = self._host(t).reshape(-1)
        r = h.new_empty(h.shape)
        self._d.all_to_all_single(r, h)
        out.copy_(r.reshape(out.shape))

    def gather(self, t, gather_list=None, dst=0):
        h = self._host(t)
        parts = [h.new_empty(h.shape) for _ in range(self.get_world_size())] if self.get_rank() == dst else None
        self._d.gather(h, parts, dst=dst)
        if parts is not None:
            for o, p in zip(gather_list, parts):
                o.copy_(p)


def msm_sharded_async(bbg, srs_local, d_scalars_ptr, n_local, d_partial, d_gathered, d_result, dist):
    """Stream-ordered variant used in the timed loop of bench.py: nothing leaves the GPU.  d_partial int64[12],
    d_gathered int64[world*12], d_result int64[12] are torch CUDA tensors; bbg must run on torch's current stream
    (bbg.set_stream) so that the RCCL all-gather is ordered after the MSM kernels and before the group sum."""
    bbg.msm_device(srs_local, d_scalars_ptr, n_local, d_partial.data_ptr())
    bbg.join()  # the reduce phase may run on the auxiliary stream: order the all-gather after it (device-side wait)
    dist.all_gather_into_tensor(d_gathered, d_partial)
    bbg.g1_sum_device(d_gathered.data_ptr(), dist.get_world_size(), d_result.data_ptr())


# ------------------------------------------------------------------------------------------------ NTT sharded across GPUs
_R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def _mont_limbs(value):
    """Plain integer -> Montgomery-form Fr as 4 uint64 limbs (host-side constant preparation only)."""
    v = (value % _R_MOD) * (1 << 256) % _R_MOD
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


class BbgNttOps:
    """Device operations of the sharded NTT on torch CUDA tensors (int64 views of n x 4 limbs)."""

    def __init__(self, bbg):
        self.bbg = bbg

    def ntt(self, d_x, log2m, op):
        self.bbg.ntt_device(d_x.data_ptr(), log2m, op)

    def scale_powers(self, d_x, count, base, start=None):
        self.bbg.scale_powers_device(d_x.data_ptr(), count, base, start)

    def root_pow(self, log2n, e, inverse):
        return self.bbg.fr_root_pow(log2n, e, inverse)

    def fr_pow(self, base, e):
        return self.bbg.fr_pow(base, e)

    def cross_dft(self, d_in, d_out, log2g, length, log2n, inverse):
        self.bbg.cross_dft_device(d_in.data_ptr(), d_out.data_ptr(), log2g, length, log2n, inverse)

    def coset_fft_shift(self, d_x, log2m, shift):
        self.bbg.ntt_device(d_x.data_ptr(), log2m, 6, 0, shift)  # BBG_COSET_FFT_WITH_GENERATOR_SHIFT


def coset_fft_split_sharded(ops, dist, x, log2n, ext):
    """The prover's 4n-point coset FFT of a polynomial with only n non-zero coefficients (large_domain.generator_size = n,
    SURVEY.md 8e row 3; reference polynomial_arithmetic.cpp:401-456 and the WASM SMALL_FFT items, work_queue.hpp:166-199):
    ext fully independent size-n coset FFTs with shifts g * w_{ext n}^k, k < ext -- NO arithmetic exchange between ranks.

    In : every rank holds the same n = 2^log2n coefficients x (int64 tensor of 4*n words; not modified).
    Out: every rank holds the ext*n interleaved evaluations, out[ext*i + k] = Y_k[i]  (what coset_fft(coeffs, small, large, ext) leaves).
    Rank g computes the cosets k = g, g + G, ... ; one all-gather of the results; the interleave is a local copy.
    ops.coset_fft_shift(d_x, log2n, shift) = coset_fft_with_generator_shift in place."""
    import torch
    G = dist.get_world_size() if dist is not None else 1
    g = dist.get_rank() if dist is not None else 0
    if ext % G:
        raise ValueError("the number of cosets must be a multiple of the world size")
    n = 1 << log2n
    log2ext = ext.bit_length() - 1
    if (1 << log2ext) != ext or x.numel() != 4 * n:
        raise ValueError("ext must be a power of two and x must hold n elements")
    per = ext // G
    local = x.new_empty((per, 4 * n))
    for j in range(per):
        k = g + G * j
        local[j].copy_(x)
        ops.coset_fft_shift(local[j], log2n, ops.root_pow(log2n + log2ext, k, False))
    if G > 1:
        flat = x.new_empty((G * per, 4 * n))  # rank-major concatenation along dim 0
        dist.all_gather_into_tensor(flat, local)
        gathered = flat.view(G, per, 4 * n)
    else:
        gathered = local.view(1, per, 4 * n)
    # gathered[g, j] = Y_{g + G j}  ->  out[i, k] with k = g + G j
    y = gathered.permute(1, 0, 2).reshape(ext, n, 4)  # index k = j*G + g
    return y.permute(1, 0, 2).contiguous().view(-1)


def ntt_sharded(ops, dist, x_local, log2n, inverse=False, coset_shift=None, new_like=None):
    """One size-n = 2^log2n (coset) NTT over G = world ranks (G a power of two <= 8, G^2 | n).

    In : rank g holds the residue class x_local[j] = a[g + G*j], j < m = n/G   (int64 tensor of 4*m words).
    Out: rank g holds out[t*(m/G) + q'] = A[(g*m/G + q') + m*t], q' < m/G, t < G   (same size).

    Decomposition (the reference's own precedent is the 4-way coset split of work_queue.hpp:166-199):
      A[q + m t] = sum_g w_G^(g t) * ( w_n^(g q) * NTT_m(x_g)[q] )
    i.e. a local size-m transform, a twiddle by powers of w_n^g, ONE all-to-all that moves (G-1)/G^2 of the data per
    rank (RCCL over xGMI on the GPU box), and a size-G DFT across the received chunks.  A coset transform
    (a_j -> a_j c^j first) scales the local residue class by c^g (c^G)^j.  The inverse transform uses the inverse roots,
    the local iNTT's 1/m and a final 1/G folded into the twiddle step.
    """
    G = dist.get_world_size() if dist is not None else 1
    g = dist.get_rank() if dist is not None else 0
    log2g = G.bit_length() - 1
    if (1 << log2g) != G or G > 8:
        raise ValueError("world size must be a power of two <= 8")
    n = 1 << log2n
    m = n // G
    if m % G or x_local.numel() != 4 * m:
        raise ValueError("need G^2 | n and a local slice of n/G elements")
    log2m = log2n - log2g
    lenq = m // G
    if coset_shift is not None:  # a[g + G j] * c^(g + G j)
        ops.scale_powers(x_local, m, ops.fr_pow(coset_shift, G), ops.fr_pow(coset_shift, g))
    ops.ntt(x_local, log2m, 1 if inverse else 0)
    start = None
    if inverse and G > 1:
        start = _mont_limbs(pow(G, -1, _R_MOD))
    if G > 1 or start is not None:
        ops.scale_powers(x_local, m, ops.root_pow(log2n, g, inverse), start)  # * w_n^(+-g q) [* 1/G]
    if G == 1:
        return x_local
    recv = new_like(x_local) if new_like is not None else x_local.clone()
    dist.all_to_all_single(recv, x_local)  # chunk s of recv = Z_s[g*lenq : (g+1)*lenq]
    out = new_like(x_local) if new_like is not None else x_local.clone()
    ops.cross_dft(recv, out, log2g, lenq, log2n, inverse)
    return out


def gather_natural_order(dist, out_local, log2n):
    """Collects the outputs of ntt_sharded on rank 0 and puts them into natural order: rank g holds out[t*(m/G) + q] = A[(g*m/G + q) + m*t]
    (m = n/G).  Returns the (n, 4) uint64 array on rank 0, None elsewhere.  Used by bench.py's self-checking config 5 and by the gloo test."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    n = 1 << log2n
    m = n // world
    lenq = m // world
    if world > 1:
        parts = [torch.empty_like(out_local) for _ in range(world)] if rank == 0 else None
        dist.gather(out_local, parts, dst=0)
    else:
        parts = [out_local]
    if rank != 0:
        return None
    nat = np.empty((n, 4), dtype=np.uint64)
    for g, part in enumerate(parts):
        o = part.cpu().numpy().view(np.uint64).reshape(world, lenq, 4)
        for t in range(world):
            nat[t * m + g * lenq: t * m + (g + 1) * lenq] = o[t]
    return nat
