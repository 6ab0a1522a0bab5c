#!/bin/bash
# Register / spill / scratch table of single kernel instantiations, in seconds (no GPU): the device pass of a tiny translation
# unit that includes the kernel headers with -DIRIS_KERNELS_ONLY (host launch code hidden) and explicitly instantiates what is
# asked for.  usage: tools/kernel_probe.sh 'mrf_conv_mfma_f32_kernel<1, 4, 4, 64, 2, 3, 7, 11, false, 0, 2, true>(const ConvLaunch)' \
#                                          'convt_mfma_f32_kernel<2, 2, 1, 4, 2, 3>(const ConvtLaunch)' ... [-- extra hipcc flags]
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd); TMP=$(mktemp -d)
{ echo '#include <hip/hip_runtime.h>'; echo '#include "generator_internal.h"'; echo '#include "conv_mfma_f32.h"'
  echo '#include "mrf_conv_mfma_f32.h"'; echo '#include "convt_mfma_f32.h"'; echo 'using namespace iris;'; } > $TMP/probe.hip
FLAGS=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; FLAGS=("$@"); break; fi
  echo "template __global__ void iris::$1;" >> $TMP/probe.hip; shift
done
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -DIRIS_KERNELS_ONLY --cuda-device-only -S \
    -I$REPO/iris-tts_amd/csrc "${FLAGS[@]}" -o $TMP/probe.s $TMP/probe.hip
python3 - $TMP/probe.s <<'PY'
import re, subprocess, sys
s = open(sys.argv[1]).read()
for blk in re.findall(r'  - \.agpr_count:.*?\.wavefront_size:', s, re.S):
    name = subprocess.run(['c++filt', re.search(r'\.name:\s+(\S+)', blk).group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(iris::\w+\)$', '', name.replace('void iris::', ''))
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    print(f"{name:72s} vgpr {g('vgpr_count'):>3} vspill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>3} sspill {g('sgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} B")
PY
rm -rf $TMP
