#!/bin/bash
# round 3, call n: the short-input instantiation of the generic conv kernel (next C_in chunk prefetched) -- parity, bit identity, timing
O=gpurun_out/r03n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.txt 2>&1; echo "pytest parity rc=$?" | tee -a $O/summary.txt; tail -2 $O/pytest_parity.txt
timeout -k 10 600 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_noshortpf.so > $O/bitwise_sweep.txt 2>&1; echo "bitwise sweep rc=$?" | tee -a $O/summary.txt; tail -1 $O/bitwise_sweep.txt
for i in 1 2; do for V in release noshortpf; do
  if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
  for T in 64 100 200 282; do
    echo "== $V T=$T: $(python tools/per_launch.py 1 $T 2>/dev/null | grep -E 'upsample' | awk '{printf "%s ", $7}') | $(python bench.py --frames $T --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step %.3f ms' % d['ms_per_step'])")" | tee -a $O/short_inputs.txt
  done
done; done
