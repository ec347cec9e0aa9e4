#!/usr/bin/env python3
"""Round 5: isolated NTT timing of ONE library build (BBG_LIB_PATH selects an A/B build from build_ab/, scripts/ab/ntt_variant.sh): HIP events
around bursts of in-place forward transforms, best of 5 bursts.  No result check -- several of the builds are timing experiments whose
results are wrong by construction (exchange or global traffic removed).  Usage: r05_ntt_time.py [log2n ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

pkg = ge.load_package()
bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
bbg.set_option("ntt_limbs29", int(os.environ.get("R05_LIMBS29", "1")))
if os.environ.get("R05_BIG_TILE"):
    bbg.set_option("ntt_big_tile", int(os.environ["R05_BIG_TILE"]))
check = "--check" in sys.argv  # forward + coset transforms of the size against the REFERENCE digests (tests/golden: committed fixture data)
sizes = [int(a) for a in sys.argv[1:] if a != "--check"] or [20, 22]


def digests_ok(lg):
    import hashlib
    import json
    gd = os.path.join(ROOT, "tests", "golden")
    recs = [(900 + lg, r["op"], r["sha256"]) for r in json.load(open(os.path.join(gd, "ntt_large.json")))["ntt"] if r["log2n"] == lg]
    if not recs:
        recs = [(r["seed"], r["op"], r["sha256"]) for r in json.load(open(os.path.join(gd, "golden.json")))["ntt"]
                if r["log2n"] == lg and r["op"] < 4 and r["generator_size"] == 0]
    ok = bool(recs)
    for seed, op, want in recs:
        w = torch.from_numpy(pkg.synthetic_scalars(seed, 1 << lg).view(np.int64).reshape(-1)).cuda()
        bbg.ntt_device(w.data_ptr(), lg, op)
        torch.cuda.synchronize()
        got = pkg.fr_reduce_once(w.cpu().numpy().view(np.uint64).reshape(-1, 4))
        ok = ok and hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == want
    return ok

tag = os.path.basename(os.environ.get("BBG_LIB_PATH", "libbbg.so"))
row = []
for lg in sizes:
    n = 1 << lg
    a = torch.from_numpy(pkg.synthetic_scalars(11, n).view(np.int64).reshape(-1)).cuda()
    for _ in range(3):
        bbg.ntt_device(a.data_ptr(), lg, 0)
    torch.cuda.synchronize()
    best = 1e9
    reps = 50 if lg <= 21 else 12
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            bbg.ntt_device(a.data_ptr(), lg, 0)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    row.append("2^%d %.4f ms%s" % (lg, best, (" digests " + ("OK" if digests_ok(lg) else "WRONG")) if check else ""))
print("%-28s %s" % (tag, "  ".join(row)), flush=True)
