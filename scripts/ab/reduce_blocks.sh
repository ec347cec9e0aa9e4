for rb in 0 256 512 1024 0 256 512 1024; do
  python bench.py --no-cpu-baseline --no-config5 --no-prover-shaped --no-sweeps --reduce-blocks $rb 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rb=$rb', d['value'], d['ms_per_step'], d['extra']['msm_phase_ms'], d['extra']['ntt_ms'], d['extra']['timed_blocks_ms'])"
done
