"""world_size-2 gloo test of the multi-GPU decomposition (aztec-2.0_amd/parallel.py) on CPU: the point-range sharding,
the all_gather of 96-byte partials and the final group sum reproduce the single-process MSM.  The per-rank "device
MSM" is emulated with the oracle (this is a CPU test of the host logic; the GPU kernels are covered by -m gpu)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["BBG_ROOT"])
import torch.distributed as dist
import __graft_entry__ as ge
from oracle.oracle import Oracle
pkg = ge.load_package()
import importlib
par = importlib.import_module("aztec_amd.parallel")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
O = Oracle()
n = 1000
pts = O.srs_hashed(0xBB254, n)
sc = pkg.synthetic_scalars(4, n)
start, count = par.shard_range(n, rank, world)
aff = O.pippenger(sc[start:start + count], pts[start:start + count])
one = O.to_mont(1, np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]
jac = np.concatenate([aff, one]) if (int(aff[3]) >> 63) == 0 else np.concatenate([aff, np.zeros(4, dtype=np.uint64)])
parts = par.all_gather_partials(jac, dist)
assert parts.shape == (world, 12)
total = O.g1_sum(parts)
whole = O.pippenger(sc, pts)
assert np.array_equal(total, whole), "sharded MSM != whole MSM"
covered = sum(par.shard_range(n, r, world)[1] for r in range(world))
assert covered == n

# the software-pipelined sharded MSM used by bench.py, with the device operations emulated by the oracle
import torch
def to_jac(aff):
    return np.concatenate([aff, one]) if (int(aff[3]) >> 63) == 0 else np.concatenate([aff, np.zeros(4, dtype=np.uint64)])
class CpuOps:
    def msm(self, scal, cnt, out):
        s_np = scal.numpy().view(np.uint64).reshape(-1, 4)
        out.copy_(torch.from_numpy(to_jac(O.pippenger(s_np[:cnt], pts[start:start + cnt])).view(np.int64).copy()))
    def join(self, lag):
        pass
    def g1_sum(self, gathered, cnt, out):
        g = gathered.numpy().view(np.uint64).reshape(cnt, 12)
        out.copy_(torch.from_numpy(to_jac(O.g1_sum(g)).view(np.int64).copy()))
pipe = par.ShardedMsmPipeline(CpuOps(), dist, lambda k: torch.zeros(k, dtype=torch.int64))
batches = [pkg.synthetic_scalars(100 + b, n) for b in range(3)]
got = []
for b in range(3):
    pipe.submit(torch.from_numpy(batches[b][start:start + count].view(np.int64).copy()), count)
    if b >= 1:
        got.append(pipe.results[(b - 1) & 1].numpy().view(np.uint64).copy())
got.append(pipe.flush().numpy().view(np.uint64).copy())
for b in range(3):
    assert np.array_equal(O.jac_to_affine(got[b]), O.pippenger(batches[b], pts)), ("pipeline", b)

# the same pipeline exchanging two MSMs' partials per all-gather (depth 2), five MSMs: two full batches and a flushed remainder
pipe2 = par.ShardedMsmPipeline(CpuOps(), dist, lambda k: torch.zeros(k, dtype=torch.int64), depth=2)
batches5 = [pkg.synthetic_scalars(200 + b, n) for b in range(5)]
seen = {}
for b in range(5):
    pipe2.submit(torch.from_numpy(batches5[b][start:start + count].view(np.int64).copy()), count)
    for j in range(pipe2.finished):  # results stay valid until their ring slot is reused (2 * depth MSMs later)
        if j not in seen and b - j < 4:
            seen[j] = pipe2.results[j % 4].numpy().view(np.uint64).copy()
last = pipe2.flush().numpy().view(np.uint64).copy()
for j in range(pipe2.finished):
    if j not in seen:
        seen[j] = pipe2.results[j % 4].numpy().view(np.uint64).copy()
assert pipe2.finished == 5 and np.array_equal(last, seen[4])
for b in range(5):
    assert np.array_equal(O.jac_to_affine(seen[b]), O.pippenger(batches5[b], pts)), ("pipeline depth 2", b)

# NTT sharded by residue class with one all-to-all: device ops emulated with the oracle
class CpuNttOps:
    def _np(self, t): return t.numpy().view(np.uint64).reshape(-1, 4)
    def ntt(self, t, log2m, op):
        a = self._np(t); a[:] = O.ntt(a.copy(), op)
    def scale_powers(self, t, cnt, base, start=None):
        a = self._np(t)
        cur = start if start is not None else O.to_mont(0, np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]
        for j in range(cnt):
            a[j] = O.fe_mul(0, a[j], cur)[0]
            cur = O.fe_mul(0, cur, base)[0]
    def _pow(self, base, e):
        acc = O.to_mont(0, np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]
        b = np.array(base, dtype=np.uint64)
        while e:
            if e & 1: acc = O.fe_mul(0, acc, b)[0]
            b = O.fe_mul(0, b, b)[0]
            e >>= 1
        return acc
    def root_pow(self, log2n, e, inverse):
        w = O.root_of_unity(log2n)
        if inverse: w = O.fe_inv(0, w)[0]
        return self._pow(w, e)
    def fr_pow(self, base, e): return self._pow(base, e)
    def coset_fft_shift(self, t, log2m, shift):
        a = self._np(t); a[:] = O.ntt(a.copy(), 6, 0, np.array(shift, dtype=np.uint64))
    def cross_dft(self, tin, tout, log2g, length, log2n, inverse):
        Gn = 1 << log2g
        a, o = self._np(tin).reshape(Gn, length, 4), self._np(tout).reshape(Gn, length, 4)
        wG = self.root_pow(log2n, (1 << log2n) // Gn, inverse)
        for t_ in range(Gn):
            acc = np.zeros((length, 4), dtype=np.uint64)
            for s_ in range(Gn):
                acc = O.fe_add(0, acc, O.fe_mul(0, a[s_], np.tile(self._pow(wG, s_ * t_), (length, 1))))
            o[t_] = acc
lg = 8
nn = 1 << lg
coeffs = pkg.synthetic_scalars(777, nn)
five = O.to_mont(0, np.array([[5, 0, 0, 0]], dtype=np.uint64))[0]
for inverse, shift, op in ((False, None, 0), (True, None, 1), (False, five, 2)):
    xl = torch.from_numpy(O.canon(0, coeffs)[rank::world].copy().view(np.int64).reshape(-1))
    res = par.ntt_sharded(CpuNttOps(), dist, xl, lg, inverse=inverse, coset_shift=shift)
    whole = O.ntt(coeffs, op)
    mm = nn // world; lenq = mm // world
    got = O.canon(0, res.numpy().view(np.uint64).reshape(-1, 4)).reshape(world, lenq, 4)
    for t_ in range(world):
        for q_ in range(lenq):
            assert np.array_equal(got[t_, q_], whole[(rank * lenq + q_) + mm * t_]), ("sharded ntt", op, t_, q_)
# what bench.py's self-checking config 5 does with the shards: the strided input generator, gather on rank 0, natural order, canonical form
import hashlib
xl = torch.from_numpy(pkg.synthetic_scalars_strided(900 + lg, nn // world, rank, world).view(np.int64).reshape(-1).copy())
assert np.array_equal(xl.numpy().view(np.uint64).reshape(-1, 4), pkg.synthetic_scalars(900 + lg, nn)[rank::world])
res = par.ntt_sharded(CpuNttOps(), dist, xl, lg, coset_shift=five)
nat = par.gather_natural_order(dist, res, lg)
if rank == 0:
    want = O.ntt(pkg.synthetic_scalars(900 + lg, nn), 2)
    assert hashlib.sha256(pkg.fr_reduce_once(nat).tobytes()).hexdigest() == hashlib.sha256(np.ascontiguousarray(want).tobytes()).hexdigest(), "config-5 style gather"
else:
    assert nat is None
# the prover's 4n coset FFT with n non-zero coefficients: ext independent size-n coset FFTs, no arithmetic exchange (8e row 3)
for lg6, ext in ((6, 4), (5, 8), (6, 2)):
    c6 = O.canon(0, pkg.synthetic_scalars(778 + ext, 1 << lg6))
    xt = torch.from_numpy(c6.copy().view(np.int64).reshape(-1))
    res = par.coset_fft_split_sharded(CpuNttOps(), dist, xt, lg6, ext)
    want = O.coset_fft_split(c6, ext)
    assert np.array_equal(O.canon(0, res.numpy().view(np.uint64).reshape(-1, 4)), O.canon(0, want)), ("coset split sharded", lg6, ext)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_msm_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BBG_ROOT=ROOT,
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=300)
        outs.append(out.decode())
    for rank, p in enumerate(procs):
        assert p.returncode == 0, outs[rank]
        assert f"rank {rank} ok" in outs[rank]


def test_shard_range_partitions():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.load_package()
    import importlib
    par = importlib.import_module("aztec_amd.parallel")
    for n in (0, 1, 7, 1 << 20, (1 << 20) + 3):
        for world in (1, 2, 4, 8):
            pos = 0
            for r in range(world):
                s, c = par.shard_range(n, r, world)
                assert s == pos and c >= 0
                pos += c
            assert pos == n
