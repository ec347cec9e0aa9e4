#!/usr/bin/env python3
"""Round 4 A/B: the quotient widgets on lazily reduced 29-bit limbs (quotient29.hip.h, option quotient_limbs29 = 1) against the 32-bit-limb
kernels (0).
  (1) the widgets one at a time through bbg_quotient_widget_device on a 4n = 2^22 domain (HIP events around each call; the call's own
      set-up kernel and the D2H of alpha_out are inside: the same for both)
  (2) bench.py's prover_shaped (config 4 on the resident prover rounds: round 4 runs permutation + fused arithmetic / range / logic +
      fixed-base) at 2^16 / 2^18 / 2^20 gates.
Writes gpurun_out/r04_widget_ab.txt."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

pkg = ge.load_package()
import bench  # noqa: E402

out_path = os.path.join(ROOT, "gpurun_out", os.environ.get("R04_WIDGET_AB_OUT", "r04_widget_ab.txt"))
os.makedirs(os.path.dirname(out_path), exist_ok=True)
out = open(out_path, "w")


def emit(line):
    print(line, file=sys.stderr, flush=True)
    out.write(line + "\n")
    out.flush()


bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
NAMES = {0: "permutation<4>", 1: "turbo arithmetic", 2: "fixed base (2 kernels)", 3: "range", 4: "logic", 5: "permutation<3>"}
QUICK = os.environ.get("R04_WIDGET_AB_QUICK") == "1"  # occupancy builds: the 2^20-gate proof only
lg = 12 if QUICK else 22
m = 1 << lg
polys = [torch.from_numpy(pkg.synthetic_scalars(7000 + k, m).view(np.int64).reshape(-1)).cuda() for k in range(21)]
ptrs = [p.data_ptr() for p in polys]
quot = torch.zeros(m * 4, dtype=torch.int64, device="cuda")
ch9 = pkg.synthetic_scalars(7100, 9)
emit(f"# (1) one widget per call, 4n = 2^{lg} points, ms (median of 9, HIP events around bbg_quotient_widget_device)")
emit("widget                      limbs32_ms  limbs29_ms  ratio")
for widget in (0, 2, 5, 1, 3, 4):
    row = []
    for limbs29 in (0, 1):
        bbg.set_option("quotient_limbs29", limbs29)
        ts = []
        for rep in range(11):
            if widget not in (0, 5):
                quot.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bbg.quotient_widget_device(widget, ptrs, lg, ch9, quot.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            if rep >= 2:
                ts.append(e0.elapsed_time(e1))
        row.append(sorted(ts)[len(ts) // 2])
    emit(f"{NAMES[widget]:26s}  {row[0]:10.3f}  {row[1]:10.3f}  {row[0] / row[1]:5.2f}")
bbg.set_option("quotient_limbs29", 1)
del polys, quot
emit("# (2) prover_shaped (bench.py, TurboPLONK-shaped proof on the resident rounds): proof and round-4 time, ms")
emit("lg   limbs29  proof_ms  round4_ms")
for lg in ((20,) if QUICK else (16, 18, 20)):
    srs = bbg.srs_synth_hashed(0xBB254, 1 << lg)
    for limbs29 in (0, 1, 0, 1):
        bbg.set_option("quotient_limbs29", limbs29)
        r = bench.prover_shaped(pkg, bbg, srs, lg, reps=7)
        emit(f"{lg:2d}   {limbs29:7d}  {r['proof_ms']:8.3f}  {r['round_ms']['round4_quotient']:9.3f}")
    bbg.set_option("quotient_limbs29", 1)
    srs.free()
out.close()
