python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msm or memory or prover or wrapped or smoke or cbind or c_binding" 2>&1 | tail -8
python tests/tools/msm_window_sweep.py 8 16 8 13 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_tiny_sweep.txt
bash scripts/ab/small_msm_timeline.sh 12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_tiny_timeline_12.txt
for lg in 10 12 14 16; do
python bench.py --log2n $lg --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline --no-config5 --no-sweeps 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('prover_shaped 2^$lg', d['extra']['prover_shaped'].get('proof_ms'), 'step ms', d['ms_per_step'])"
done
