#!/bin/bash
# usage: run.sh <tag> [extra flags]   -> /tmp/probe/<tag>.s and a one-line summary per kernel
tag=$1; shift
cd /root/repo/iris-tts_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -DIRIS_KERNELS_ONLY --cuda-device-only -S -I. "$@" -o /tmp/probe/$tag.s ${PROBE_SRC:-/tmp/probe/probe.hip} 2>/tmp/probe/$tag.err || { tail -20 /tmp/probe/$tag.err; exit 1; }
python3 - /tmp/probe/$tag.s <<'PY'
import re,sys,subprocess
s=open(sys.argv[1]).read()
for blk in re.findall(r'  - \.agpr_count:.*?\.wavefront_size:', s, re.S):
    name=re.search(r'\.name:\s+(\S+)',blk).group(1)
    name=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip().replace('void iris::','').replace('(iris::ConvLaunch)','').replace('(iris::ConvtLaunch)','')
    g=lambda k: re.search(r'\.%s:\s+(\d+)'%k,blk).group(1)
    print(f"{name:70s} vgpr {g('vgpr_count'):>3} agpr {g('agpr_count'):>3} vspill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>3} sspill {g('sgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size')}")
PY
