#!/bin/bash
# round 3, call d: persistent fp32 conv pairs with jobs drawn from a counter -- parity, then which variant wins where
set -o pipefail
O=gpurun_out/r03d
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_pair or generator or config" > $O/pytest_parity.txt 2>&1; echo "pytest parity rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_parity.txt
CFGS=("IRIS_HIFIGAN_PAIR_PF_MODE=0" "X=0" "IRIS_HIFIGAN_PAIR_PF_DYN=0" "IRIS_HIFIGAN_PAIR_PF_GRID=1" "IRIS_HIFIGAN_PAIR_PF_NONSUM=0" "IRIS_HIFIGAN_PAIR_PF_NONSUM=0 IRIS_HIFIGAN_PAIR_SUM=2" "IRIS_HIFIGAN_PAIR_SUM=0" "IRIS_HIFIGAN_PAIR_PF_MODE=0")
timeout -k 10 900 bash tools/ablate.sh "${CFGS[@]}" 2>&1 | grep -v amdgpu.ids | tee $O/ablate_1x1000.txt
BENCH_ARGS="--batch 32 --frames 500" timeout -k 10 900 bash tools/ablate.sh "${CFGS[@]}" 2>&1 | grep -v amdgpu.ids | tee $O/ablate_32x500.txt
BENCH_ARGS="--batch 1 --frames 500" timeout -k 10 600 bash tools/ablate.sh "IRIS_HIFIGAN_PAIR_PF_MODE=0" "X=0" "IRIS_HIFIGAN_PAIR_PF_NONSUM=0" 2>&1 | grep -v amdgpu.ids | tee $O/ablate_1x500.txt
