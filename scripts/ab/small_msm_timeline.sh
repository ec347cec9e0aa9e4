# kernel timeline of stand-alone small MSMs (2^12, 2^16): where the latency goes
cd /tmp && export TMPDIR=/tmp
cat > /tmp/small.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as ge, torch, time
pkg = ge.load_package(); bbg = pkg.Bbg(0); bbg.set_stream(torch.cuda.current_stream().cuda_stream)
for lg in (12, 16):
    n = 1 << lg
    srs = bbg.srs_synth_hashed(0xBB254, n)
    sc = torch.from_numpy(pkg.synthetic_scalars(7, n).view(np.int64).reshape(-1)).cuda()
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    for _ in range(12):
        bbg.msm_device(srs, sc.data_ptr(), n, out.data_ptr()); bbg.sync()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); bbg.msm_device(srs, sc.data_ptr(), n, out.data_ptr()); bbg.sync(); ts.append(time.perf_counter() - t0)
    print("n=2^%d standalone median %.3f ms" % (lg, sorted(ts)[10] * 1e3))
PY
rm -rf /tmp/kts; rocprofv3 --kernel-trace -d /tmp/kts -o s -- python /tmp/small.py 2>&1 | grep standalone
DB=$(find /tmp/kts -name "*_results.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocpd_timeline.py $DB "k_sortA_count<16>" 8 11
echo ---
python $GRAFT_REPO_ROOT/scripts/rocpd_timeline.py $DB "k_sortA_count<16>" 40 11
