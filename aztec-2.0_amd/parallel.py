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
    """Adapter: the three device operations the sharded pipeline needs, on torch CUDA tensors."""

    def __init__(self, bbg, srs):
        self.bbg, self.srs = bbg, srs

    def msm(self, d_scalars, n, d_out):
        self.bbg.msm_device(self.srs, d_scalars.data_ptr(), n, d_out.data_ptr())

    def join(self, lag):
        self.bbg.join(lag)

    def g1_sum(self, d_jacobians, count, d_out):
        self.bbg.g1_sum_device(d_jacobians.data_ptr(), count, d_out.data_ptr())


class ShardedMsmPipeline:
    """One global MSM per submit(), sharded by point range over the ranks of `dist`, software-pipelined by one call:
    while rank-local MSM i is still in its bucket-reduction phase (auxiliary stream), the 96-byte partials of MSM i-1
    are all-gathered (RCCL) and summed.  flush() completes the last one.  Everything is stream-ordered; nothing
    returns to the host.  results[i & 1] holds global result i after the corresponding finish step."""

    def __init__(self, ops, dist, new_tensor):
        self.ops, self.dist = ops, dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.partial = [new_tensor(12), new_tensor(12)]
        self.gathered = new_tensor(12 * self.world)
        self.results = [new_tensor(12), new_tensor(12)]
        self.count = 0

    def _finish(self, j, lag):
        self.ops.join(lag)  # device-side wait for the reduction of MSM j (not for the one issued after it)
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.gathered, self.partial[j & 1])
            self.ops.g1_sum(self.gathered, self.world, self.results[j & 1])
        else:
            self.results[j & 1].copy_(self.partial[j & 1])

    def submit(self, d_scalars, n):
        i = self.count
        self.ops.msm(d_scalars, n, self.partial[i & 1])
        if i >= 1:
            self._finish(i - 1, 1)
        self.count += 1

    def flush(self):
        if self.count >= 1:
            self._finish(self.count - 1, 0)
        return self.results[(self.count - 1) & 1] if self.count else None


def msm_sharded_async(bbg, srs_local, d_scalars_ptr, n_local, d_partial, d_gathered, d_result, dist):
    """Stream-ordered variant used in the timed loop of bench.py: nothing leaves the GPU.  d_partial int64[12],
    d_gathered int64[world*12], d_result int64[12] are torch CUDA tensors; bbg must run on torch's current stream
    (bbg.set_stream) so that the RCCL all-gather is ordered after the MSM kernels and before the group sum."""
    bbg.msm_device(srs_local, d_scalars_ptr, n_local, d_partial.data_ptr())
    bbg.join()  # the reduce phase may run on the auxiliary stream: order the all-gather after it (device-side wait)
    dist.all_gather_into_tensor(d_gathered, d_partial)
    bbg.g1_sum_device(d_gathered.data_ptr(), dist.get_world_size(), d_result.data_ptr())
