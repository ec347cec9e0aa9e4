#!/usr/bin/env python3
"""BASELINE.json config 5: ONE 2^24-point MSM and ONE 2^24 coset-NTT sharded across the N GPUs of a node (strong scaling).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_sharded.py \
        [--log2n 24] [--steps 5] [--warmup 1]

MSM  : point-range shards (n/N points per rank), RCCL all-gather of the N 96-byte partials + group sum.
NTT  : residue-class shards, local size-n/N coset NTT, twiddle, ONE RCCL all-to-all ((N-1)/N^2 of the data per rank),
       size-N DFT across the received chunks (aztec-2.0_amd/parallel.py::ntt_sharded).
Rank 0 prints one JSON line with both rates.  (bench.py is the driver's headline benchmark; this script reports the
strong-scaling configuration and is not run by default.)
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one JSON line on stdout: native libraries (RCCL) print to fd 1 behind Python's back, so fd 1 is pointed at stderr for the run
REAL_STDOUT = os.dup(1)
os.dup2(2, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=24)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    par = importlib.import_module("aztec_amd.parallel")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    lg, n = args.log2n, 1 << args.log2n
    bbg = pkg.Bbg(local_rank)
    bbg.set_stream(torch.cuda.current_stream().cuda_stream)
    start, count = par.shard_range(n, rank, world)
    srs = bbg.srs_synth_hashed(0xBB254 + start, count)
    d_scalars = torch.from_numpy(pkg.synthetic_scalars(0xBB254 + 3, count, start).view(np.int64)).to(dev)
    pipe = par.ShardedMsmPipeline(par.BbgOps(bbg, srs), dist, lambda k: torch.zeros(k, dtype=torch.int64, device=dev))
    m = n // world
    # residue class of a deterministic coefficient vector: element j of the class = global coefficient rank + world*j
    full_idx = rank + world * np.arange(m, dtype=np.uint64)
    coeff = pkg.inputs.splitmix64_limbs(0xBB254 + 100 + lg, 4 * n).reshape(n, 4)[full_idx] if n <= (1 << 22) else \
        pkg.synthetic_scalars(0xBB254 + 100 + lg + rank, m)
    coeff[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    d_x = torch.from_numpy(coeff.view(np.int64).reshape(-1)).to(dev)
    ops = par.BbgNttOps(bbg)
    five = np.array([5, 0, 0, 0], dtype=np.uint64)
    shift = bbg.field_op(0, 5, five.reshape(1, 4))[0]  # 5 in Montgomery form (coset generator, fr.hpp:44-59)
    bbg.ntt_prepare(lg - (world.bit_length() - 1))

    def fence():
        dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / args.steps

    def msm_step():
        pipe.count = 0
        pipe.submit(d_scalars, count)
        pipe.flush()

    work = d_x.clone()

    def ntt_step():
        work.copy_(d_x)
        par.ntt_sharded(ops, dist, work, lg, coset_shift=shift)

    t_msm = timed(msm_step)
    t_ntt = timed(ntt_step)
    if rank == 0:
        sys.stdout.flush()
        os.write(REAL_STDOUT, (json.dumps({"workload": "one 2^%d MSM + one 2^%d coset-NTT sharded over %d GPUs (strong scaling)" % (lg, lg, world),
                          "n_gpus": world, "msm_ms": round(t_msm * 1e3, 3), "msm_mscalar_per_s": round(n / t_msm / 1e6, 2),
                          "ntt_ms": round(t_ntt * 1e3, 3), "ntt_gfield_ops_per_s": round(1.5 * n * lg / t_ntt / 1e9, 2),
                          "exchange": {"msm": "all_gather %d x 96 B" % world, "ntt": "all_to_all %.1f MiB per rank" % (32.0 * m * (world - 1) / world / 2**20)}}) + "\n").encode())
    srs.free()
    bbg.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
