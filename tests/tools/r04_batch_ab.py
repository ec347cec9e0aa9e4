#!/usr/bin/env python3
"""Round 4 A/B: the independent MSMs of a prover round as ONE launch set (bbg_msm_batch_device) against one launch set each.
  (1) K = 4 MSMs of 2^lg terms: four bbg_msm_device calls (reduce phases on the auxiliary streams) against one batch; wall clock incl. the
      final join + sync, median of 15.
  (2) bench.py's prover_shaped (config 4 on the resident prover rounds) with the option prover_msm_batch = 0 (round-3 behaviour), 2, 4.
Writes gpurun_out/r04_batch_ab.txt."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

pkg = ge.load_package()
import bench  # noqa: E402  (its prover_shaped; importing it points fd 1 at stderr)

out_path = os.path.join(ROOT, "gpurun_out", "r04_batch_ab.txt")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
out = open(out_path, "w")


def emit(line):
    print(line, file=sys.stderr, flush=True)
    out.write(line + "\n")
    out.flush()


bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
bbg.set_option("msm_async_reduce", 1)
emit("# (1) four MSMs of 2^lg terms over one SRS: 4 x bbg_msm_device vs 1 x bbg_msm_batch_device (ms per group of four, median of 15)")
emit("lg   singles_ms  batch4_ms  batch2x2_ms")
res = torch.zeros(4 * 12, dtype=torch.int64, device="cuda")
for lg in (10, 12, 14, 16, 18, 20):
    n = 1 << lg
    srs = bbg.srs_synth_hashed(0xBB254, n)
    scs = [torch.from_numpy(pkg.synthetic_scalars(100 + k, n).view(np.int64).reshape(-1)).cuda() for k in range(4)]
    ptrs = [s.data_ptr() for s in scs]

    def singles():
        for k in range(4):
            bbg.msm_device(srs, ptrs[k], n, res.data_ptr() + 96 * k)

    def batch4():
        bbg.msm_batch_device(srs, ptrs, [n] * 4, res.data_ptr())

    def batch2x2():
        bbg.msm_batch_device(srs, ptrs[:2], [n] * 2, res.data_ptr())
        bbg.msm_batch_device(srs, ptrs[2:], [n] * 2, res.data_ptr() + 192)

    row = []
    ref = None
    for fn in (singles, batch4, batch2x2):
        for _ in range(3):
            fn()
        bbg.join(); bbg.sync()
        got = bbg.g1_normalize(res.cpu().numpy().view(np.uint64).reshape(4, 12))
        if ref is None:
            ref = got
        assert np.array_equal(ref, got), (lg, fn.__name__)
        ts = []
        for _ in range(15):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            bbg.join(); bbg.sync()
            ts.append(time.perf_counter() - t0)
        row.append(sorted(ts)[len(ts) // 2] * 1e3)
    emit(f"{lg:2d}   {row[0]:10.3f}  {row[1]:9.3f}  {row[2]:11.3f}")
    srs.free()

emit("# (2) prover_shaped (bench.py, TurboPLONK-shaped proof on the resident rounds), option prover_msm_batch")
emit("lg   batch  proof_ms  rounds_ms")
for lg in (12, 14, 16, 18, 20):
    srs = bbg.srs_synth_hashed(0xBB254, 1 << lg)
    for batch in (0, 2, 4):
        bbg.set_option("prover_msm_batch", batch)
        r = bench.prover_shaped(pkg, bbg, srs, lg, reps=7)
        emit(f"{lg:2d}   {batch:5d}  {r['proof_ms']:8.3f}  {list(r['round_ms'].values())}")
    bbg.set_option("prover_msm_batch", 4)
    srs.free()
out.close()
