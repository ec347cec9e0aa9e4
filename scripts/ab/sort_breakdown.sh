# per-kernel durations of the MSM at 2^24 for 20- and 22-bit windows (rocprofv3 kernel trace of tests/tools/msm_window_sweep.py)
cd /tmp && export TMPDIR=/tmp
for lg in 24; do
rm -rf /tmp/kt$lg; rocprofv3 --kernel-trace -d /tmp/kt$lg -o ph -- python $GRAFT_REPO_ROOT/tests/tools/msm_window_sweep.py $lg $lg 22 > /tmp/ph$lg.log 2>&1
echo "== 2^$lg"; tail -2 /tmp/ph$lg.log; python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py kernels $(find /tmp/kt$lg -name "*_results.db" | head -1) | cut -c1-132 | grep -E "sort|accum|combine|rowcol|planes|kernel "
done
