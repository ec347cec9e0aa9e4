# quad products A/B: small MSM sweep + prover_shaped at 2^12 / 2^16 / 2^20 with both libraries, then the MSM / quad tests on the new one
for lib in build_ab/libbbg_quadmul32.so aztec-2.0_amd/csrc/libbbg.so build_ab/libbbg_quadmul32.so aztec-2.0_amd/csrc/libbbg.so; do
  echo "== $lib"
  BBG_LIB_PATH=$lib python tests/tools/msm_window_sweep.py 12 16 13 16 2>&1 | grep -v amdgpu.ids | awk 'NR==1 || ($1==12 && $2==13) || ($1==14 && $2==13) || ($1==16 && $2==16)'
  for lg in 12 16; do
    BBG_LIB_PATH=$lib python bench.py --log2n $lg --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline --no-config5 --no-sweeps 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('prover_shaped 2^$lg', d['extra']['prover_shaped'].get('proof_ms'), 'step ms', d['ms_per_step'])"
  done
done
BBG_LIB_PATH=build_ab/libbbg_quadmul32.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config5 --no-sweeps 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('quadmul32: prover_shaped 2^20', d['extra']['prover_shaped'].get('proof_ms'), 'step ms', d['ms_per_step'], d['extra']['msm_phase_ms'])"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config5 --no-sweeps 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('quadmul29: prover_shaped 2^20', d['extra']['prover_shaped'].get('proof_ms'), 'step ms', d['ms_per_step'], d['extra']['msm_phase_ms'])"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msm or quad or g1_sum or prover or wrapped" 2>&1 | tail -5
