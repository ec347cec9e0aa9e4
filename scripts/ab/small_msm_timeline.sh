#!/bin/bash
# kernel timeline of stand-alone MSMs at a small size: bash scripts/ab/small_msm_timeline.sh [log2n]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LG=${1:-16}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/small_msm.py <<PY
import sys, numpy as np
sys.path.insert(0, "$ROOT")
import __graft_entry__ as ge, torch
pkg = ge.load_package(); bbg = pkg.Bbg(0)
bbg.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << $LG
srs = bbg.srs_synth_hashed(0xBB254, n)
d = torch.from_numpy(pkg.synthetic_scalars(7, n).view(np.int64).reshape(-1)).cuda()
out = torch.zeros(12, dtype=torch.int64, device="cuda")
import time
for _ in range(20):
    bbg.msm_device(srs, d.data_ptr(), n, out.data_ptr()); bbg.sync()
t0 = time.perf_counter()
for _ in range(20):
    bbg.msm_device(srs, d.data_ptr(), n, out.data_ptr()); bbg.sync()
print("standalone ms", (time.perf_counter() - t0) / 20 * 1e3)
PY
rm -rf /tmp/kts && rocprofv3 --kernel-trace -d /tmp/kts -o t -- python /tmp/small_msm.py 2>&1 | grep standalone
python $ROOT/scripts/rocpd_timeline.py $(find /tmp/kts -name "*_results.db" | head -1) ${2:-k_sortA_count} 30 16
