#!/usr/bin/env python3
"""A/B helper: bench.py's prover_shaped at 2^lg gates with the library BBG_LIB_PATH names; prints proof and round times (ms)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

pkg = ge.load_package()
import bench  # noqa: E402

bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
# OPT=key=value[,key=value] sets library options first (e.g. OPT=ntt_limbs29=1)
for kv in filter(None, os.environ.get("OPT", "").split(",")):
    k, v = kv.split("=")
    bbg.set_option(k, int(v))
for lg in [int(a) for a in sys.argv[1:]] or [20]:
    srs = bbg.srs_synth_hashed(0xBB254, 1 << lg)
    for _ in range(2):
        r = bench.prover_shaped(pkg, bbg, srs, lg, reps=7)
        print(lg, r["proof_ms"], list(r["round_ms"].values()), file=sys.stderr, flush=True)
    srs.free()
