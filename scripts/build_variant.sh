#!/bin/bash
# builds ab/v_<name>.so from the tree's sources with extra -D flags: scripts/build_variant.sh name [-DK4_X=1 ...]
NAME=$1; shift
mkdir -p ab /tmp/w/bv_$NAME
cd /tmp/w/bv_$NAME && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage --save-temps "$@" /root/repo/k4os/compression/lz4_amd/csrc/k4lz4_capi.hip -o /root/repo/ab/v_$NAME.so 2> remarks.txt
python3 - <<PY
import re
txt=open('/tmp/w/bv_$NAME/remarks.txt').read()
for b in re.split(r'(?=remark: [^\n]*Function Name)', txt):
    m=re.search(r'Function Name: (\S+)', b)
    if not m or not re.search(r'encode_fast_kernel|encode_fast_gtab|decode_pair_kernel|pickle_kernel|hc_parse_kernel', m[1]): continue
    g=lambda k: (re.search(k+r': (\d+)', b) or [None,'?'])[1]
    print('$NAME %-44s SGPR %s VGPR %s scratch %s occ %s sspill %s'%(m[1][6:50], g('TotalSGPRs'), g(' VGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g('SGPRs Spill')))
if 'error' in txt: print(txt[-3000:])
PY
