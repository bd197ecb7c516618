#!/bin/bash
# Spill check of the row-stream kernels (hipcc cross-compiles without a GPU): scratch operations per instantiation and how many of them sit inside loops.
# usage (from crnn-ocr-lite_amd/csrc): ../../scripts/check_loop_spills.sh dwconv_bwd_stream.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -S --cuda-device-only $1 -o /tmp/isa/t.s 2>/dev/null
python3 - <<'E'
import re
s=open('/tmp/isa/t.s').read()
for m in re.finditer(r'^(_ZN12_GLOBAL__N_1\d+dw_\w+_stream_kernel\w+):', s, re.M):
    i=m.start(); j=s.index('.Lfunc_end', i); b=s[i:j]
    inloop=0; tot=0; cur=False
    for l in b.split('\n'):
        if l.startswith('.LBB'): cur = 'in Loop' in l or 'Loop Header' in l
        if 'scratch_' in l:
            tot+=1
            if cur: inloop+=1
    vg=re.search(r'; NumVgprs: (\d+)', s[j:j+3000]).group(1)
    print(m.group(1)[18:75], 'vgpr', vg, 'scratch ops', tot, 'in loops', inloop)
E
