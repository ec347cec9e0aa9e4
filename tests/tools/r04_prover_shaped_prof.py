#!/usr/bin/env python3
"""Workload for a kernel trace of the resident prover rounds at a small size: bench.py's prover_shaped at 2^lg (default 16), 8 proofs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

pkg = ge.load_package()
import bench  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
srs = bbg.srs_synth_hashed(0xBB254, 1 << lg)
r = bench.prover_shaped(pkg, bbg, srs, lg, reps=7)
print("prover_shaped", lg, r["proof_ms"], r["round_ms"], file=sys.stderr)
