#!/bin/bash
# Round-4 experiment (needs CARTNET_BUILD_EXPERIMENTAL=1 python -m cartnet_amd.build first):
# same box, interleaved: correctness first (GEMM / model / accuracy tests with the switch on), then every edge-sized
# form alone, then (STEP=1) the training step.
set -o pipefail
mkdir -p gpurun_out
if [ "${TESTS:-1}" == "1" ]; then
CARTNET_Q=1 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_accuracy.py -x -q -m gpu > gpurun_out/q_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/q_tests.log
fi
for v in 0 1 0 1; do
  echo "== CARTNET_Q=$v"
  CARTNET_Q=$v timeout -k 10 300 python tools/bench_gemm_forms.py 0 2>/dev/null | grep -v "thin\|enc dW\|dW:\|dW1e"
done > gpurun_out/q_forms.log 2>&1
cat gpurun_out/q_forms.log
if [ -f cartnet_amd/libcartnet_hip_stamp.so ]; then
  CARTNET_LIB=$PWD/cartnet_amd/libcartnet_hip_stamp.so CARTNET_Q=1 python tools/experiments/exp_phases.py 256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ph_q.txt
fi
if [ "${STEP:-0}" == "1" ]; then
for v in 0 1 2 0 1 2; do
  echo "== CARTNET_Q=$v"
  CARTNET_Q=$v timeout -k 10 300 python bench.py --no-subconfigs --steps 30 --warmup 10 --no-x3-pass --no-recipe-pass --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['calibration']['avg_launch_us'])"
done 2>&1 | tee gpurun_out/q_step.log
fi
