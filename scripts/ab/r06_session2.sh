set -x
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wrapped or shim or above_2_24 or bench or config5_over_rccl or prover_keeps or native" 2>&1 | tail -15
python tests/tools/msm_window_sweep.py 10 17 13 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_small_sweep.txt
python tests/tools/msm_window_sweep.py 20 20 17 19 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_w20_sweep.txt
bash scripts/ab/small_msm_timeline.sh 12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_small_timeline_12.txt
bash scripts/ab/small_msm_timeline.sh 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_small_timeline_16.txt
python tests/tools/host_msm_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_host_msm_ab.txt
