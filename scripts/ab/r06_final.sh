# round 6 final measurements (one session): GPU suite with durations, the profile set, the default bench line, the small-path threshold check
python -m pytest tests/ -q -m gpu --durations=12 2>&1 | tail -22 > gpurun_out/r06_gpu_suite.txt
bash scripts/refresh_profiles.sh v2 r06 $1 > gpurun_out/r06_refresh.log 2>&1
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_stderr.txt
for w in 0 8; do
python bench.py --log2n 13 --msm-window $w --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline --no-config5 --no-sweeps 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('prover_shaped 2^13 window $w', d['extra']['prover_shaped'].get('proof_ms'), 'step ms', d['ms_per_step'])"
done > gpurun_out/r06_tiny_threshold.txt 2>&1
