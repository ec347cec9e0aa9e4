"""gloo tests (world 2, 4 and 8) of the multi-GPU decomposition (aztec-2.0_amd/parallel.py) on CPU: the point-range sharding,
the all_gather of 96-byte partials and the final group sum reproduce the single-process MSM; the residue-class NTT with its ONE
all-to-all, the natural-order gather and the exchange-free 4n coset split reproduce the single-process transforms -- at exactly the
shapes `bench.py --gpus N` runs (ShardedMsmPipeline at depth 4 with a side stream, >= 9 MSMs; ntt_sharded fft / ifft / coset;
gather_natural_order; coset_fft_split_sharded).  The per-rank "device" operations are emulated with the oracle (this is a CPU test
of the host logic and the index arithmetic; the GPU kernels are covered by -m gpu), the stream / event API by a recording stand-in
whose log is checked against the ordering protocol."""
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

# depth 2 (two MSMs' partials per all-gather): three submits, flush (a partial batch), three more submits on the SAME pipeline without
# reset(), flush -- the ring must not wrap inside a batch and no MSM may be dropped (round-3 advisor finding)
def run_pipeline(pipe, seeds, flush_after=()):
    batches = [pkg.synthetic_scalars(s, n) for s in seeds]
    slots, snap = [], {}
    def collect(upto):  # results stay valid until their ring slot is reused (2 * depth MSMs later): copy them as soon as they are issued
        for j in range(upto):
            if j not in snap:
                snap[j] = pipe.results[slots[j]].numpy().view(np.uint64).copy()
    for b, sc_b in enumerate(batches):
        before = pipe.finished
        slots.append(pipe.submit(torch.from_numpy(sc_b[start:start + count].view(np.int64).copy()), count))
        if pipe.finished != before:  # a whole batch was finished behind this submit: everything but the MSM just issued
            collect(b)
        if b + 1 in flush_after:
            last = pipe.flush()
            collect(b + 1)
            assert np.array_equal(last.numpy().view(np.uint64), snap[b])
    last = pipe.flush()
    collect(len(batches))
    assert np.array_equal(last.numpy().view(np.uint64), snap[len(batches) - 1])
    for b, sc_b in enumerate(batches):
        assert np.array_equal(O.jac_to_affine(snap[b]), O.pippenger(sc_b, pts)), ("pipeline", pipe.depth, b)
    assert pipe.submitted == len(batches)
run_pipeline(par.ShardedMsmPipeline(CpuOps(), dist, lambda k: torch.zeros(k, dtype=torch.int64), depth=2), range(200, 206), flush_after=(3,))
run_pipeline(par.ShardedMsmPipeline(CpuOps(), dist, lambda k: torch.zeros(k, dtype=torch.int64), depth=2), range(210, 215))

# depth 4 WITH a side stream -- the configuration of bench.py --gpus N -- and ten MSMs (two full batches + a remainder of two), then a
# second timed block on the same pipeline after reset().  torch.cuda is replaced by a recording stand-in: operations execute at once (CPU),
# the log is checked against the protocol the streams must follow.
class Log(list):
    pass
LOG = Log()
class FakeStream:
    def __init__(self, name): self.name = name
    def wait_event(self, ev): LOG.append(("wait_event", self.name, ev.name, ev.recorded_on))
    def wait_stream(self, other): LOG.append(("wait_stream", self.name, other.name))
class FakeEvent:
    count = 0
    def __init__(self):
        self.name = "ev%d" % FakeEvent.count; FakeEvent.count += 1; self.recorded_on = None
    def record(self, stream): self.recorded_on = stream.name; LOG.append(("record", stream.name, self.name))
class FakeCuda:
    Event = FakeEvent
    def __init__(self): self.main = FakeStream("main"); self.cur = self.main
    def current_stream(self): return self.cur
    def stream(self, s):
        outer = self
        class Ctx:
            def __enter__(self_): self_.prev = outer.cur; outer.cur = s
            def __exit__(self_, *a): outer.cur = self_.prev
        return Ctx()
fake = FakeCuda()
class CpuOpsSide(CpuOps):
    bbg_side = object()
    def msm(self, scal, cnt, out):
        LOG.append(("msm", fake.cur.name)); CpuOps.msm(self, scal, cnt, out)
    def join(self, lag): LOG.append(("join", fake.cur.name, lag))
    def g1_sum(self, gathered, cnt, out):
        LOG.append(("g1_sum", fake.cur.name)); CpuOps.g1_sum(self, gathered, cnt, out)
side = FakeStream("side")
pipe4 = par.ShardedMsmPipeline(CpuOpsSide(), dist, lambda k: torch.zeros(k, dtype=torch.int64), side_stream=side, depth=4, cuda=fake)
run_pipeline(pipe4, range(300, 310))
assert pipe4.count == pipe4.finished == 12  # the flushed remainder ended its batch
# protocol: every local MSM is issued on the main stream, every group sum on the side stream, each exchange waits (on the side stream) for an
# event recorded on the main stream after a join, and the main stream ends by waiting for the side stream
assert all(e[1] == "main" for e in LOG if e[0] == "msm") and sum(e[0] == "msm" for e in LOG) == 10
assert all(e[1] == "side" for e in LOG if e[0] == "g1_sum") and sum(e[0] == "g1_sum" for e in LOG) == 10
assert LOG[-1] == ("wait_stream", "main", "side")
for i, e in enumerate(LOG):
    if e[0] == "g1_sum":  # the nearest preceding wait on the side stream is for an event recorded on main
        waits = [w for w in LOG[:i] if w[0] == "wait_event" and w[1] == "side"]
        assert waits and waits[-1][3] == "main", ("side stream ran ahead of the local MSMs", i)
# half h of the ring is overwritten by MSMs 8, 9 (h = 0): the main stream waited for the side stream's ev_done of that half first
idx_msm = [i for i, e in enumerate(LOG) if e[0] == "msm"]
assert any(w[0] == "wait_event" and w[1] == "main" and w[3] == "side" for w in LOG[idx_msm[7]:idx_msm[8]]), "ring half reused without waiting for its exchange"
del LOG[:]
pipe4.reset()
run_pipeline(pipe4, range(320, 329))  # nine: two full batches + one
try:
    par.ShardedMsmPipeline(CpuOps(), dist, lambda k: torch.zeros(k, dtype=torch.int64), side_stream=side, depth=4, cuda=fake)
    CpuOps.bbg_side = None
    par.ShardedMsmPipeline(CpuOps(), dist, lambda k: torch.zeros(k, dtype=torch.int64), side_stream=side, depth=4, cuda=fake)
    raise AssertionError("a side stream without a side context must be refused")
except ValueError:
    pass

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
    if ext % world:
        continue  # the cosets are dealt out whole: ext must be a multiple of the world size (world 8: ext = 8 only)
    c6 = O.canon(0, pkg.synthetic_scalars(778 + ext, 1 << lg6))
    xt = torch.from_numpy(c6.copy().view(np.int64).reshape(-1))
    res = par.coset_fft_split_sharded(CpuNttOps(), dist, xt, lg6, ext)
    want = O.coset_fft_split(c6, ext)
    assert np.array_equal(O.canon(0, res.numpy().view(np.uint64).reshape(-1, 4)), O.canon(0, want)), ("coset split sharded", lg6, ext)
# the REHEARSAL carrier of `bench.py --gpus N` with all ranks on one device (BBG_DIST_ONE_DEVICE=1): parallel.HostStagedDist stages every payload
# through the host and runs the gloo collective -- every call bench.py makes through it, on the same shapes, must give the same results
hs = par.HostStagedDist(dist)
assert hs.get_world_size() == world and hs.get_rank() == rank
tmax = torch.tensor([float(rank)], dtype=torch.float64)
hs.all_reduce(tmax, op=hs.ReduceOp.MAX)
assert float(tmax.item()) == world - 1
run_pipeline(par.ShardedMsmPipeline(CpuOps(), hs, lambda k: torch.zeros(k, dtype=torch.int64), depth=4), range(400, 406))
xl = torch.from_numpy(pkg.synthetic_scalars_strided(900 + lg, nn // world, rank, world).view(np.int64).reshape(-1).copy())
res = par.ntt_sharded(CpuNttOps(), hs, xl, lg, coset_shift=five)
nat = par.gather_natural_order(hs, res, lg)
if rank == 0:
    want = O.ntt(pkg.synthetic_scalars(900 + lg, nn), 2)
    assert np.array_equal(pkg.fr_reduce_once(nat), np.ascontiguousarray(want)), "host-staged config-5 gather"
if 8 % world == 0:
    c6 = O.canon(0, pkg.synthetic_scalars(786, 1 << 5))
    res = par.coset_fft_split_sharded(CpuNttOps(), hs, torch.from_numpy(c6.copy().view(np.int64).reshape(-1)), 5, 8)
    assert np.array_equal(O.canon(0, res.numpy().view(np.uint64).reshape(-1, 4)), O.canon(0, O.coset_fft_split(c6, 8))), "host-staged coset split"
parts_h = [torch.zeros(12, dtype=torch.int64) for _ in range(world)]
hs.all_gather(parts_h, torch.full((12,), rank, dtype=torch.int64))
assert all(int(parts_h[r][0]) == r for r in range(world))
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


def _run_world(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BBG_ROOT=ROOT,
                   OMP_NUM_THREADS="1" if world > 2 else "2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=600)
            outs.append(out.decode())
    finally:
        for p in procs:  # the exact processes started above, never a pattern
            if p.poll() is None:
                p.kill()
    for rank, p in enumerate(procs):
        assert p.returncode == 0, outs[rank]
        assert f"rank {rank} ok" in outs[rank]


def test_sharded_msm_gloo_world2(tmp_path):
    _run_world(tmp_path, 2)


def test_sharded_paths_gloo_world4(tmp_path):
    """The shapes of `bench.py --gpus 4` (reference precedent for the split: c_bind.cpp:31-46, work_queue.hpp:166-199)."""
    _run_world(tmp_path, 4)


def test_sharded_paths_gloo_world8(tmp_path):
    """The shapes of `bench.py --gpus 8`: G = 8 residue classes (G^2 | n), eight cosets dealt out one per rank, depth-4 pipeline."""
    _run_world(tmp_path, 8)


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
