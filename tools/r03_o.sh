#!/bin/bash
# round 3, call o: weight fragments of the generic fp32 conv kernel four groups ahead instead of two -- parity, bit identity, timing
O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_postnet.py tests/test_pipeline.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -2 $O/pytest.txt
timeout -k 10 600 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_ring2.so > $O/bitwise_sweep.txt 2>&1; echo "bitwise sweep rc=$?" | tee -a $O/summary.txt; tail -1 $O/bitwise_sweep.txt
for i in 1 2; do for V in release ring2; do
  if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
  for T in 64 100 282 1000; do
    echo "== $V T=$T: $(python tools/per_launch.py 1 $T 2>/dev/null | grep -E 'upsample|conv_pre' | awk '{printf "%s ", $7}') | $(python bench.py --frames $T --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step %.3f ms' % d['ms_per_step'])")" | tee -a $O/conv_weight_ring.txt
  done
done; done
