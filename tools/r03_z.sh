#!/bin/bash
# round 3, final measurement pass: full GPU suite, then tools/final_profiles.sh (bench lines, kernel stats, HBM traffic)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03zz_pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"
tail -2 gpurun_out/r03zz_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03zz_smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r03zz_smoke.txt
timeout -k 10 1500 bash tools/final_profiles.sh r03zz; echo "final_profiles rc=$?"
