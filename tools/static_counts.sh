#!/bin/bash
# Static instruction counts of the step kernel instantiations in the current sources (diagnostic; hipcc only, no GPU):
#   bash tools/static_counts.sh [extra hipcc flags]
cd "$(dirname "$0")/../everglades-ai-wargame_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" -S --cuda-device-only -o /tmp/evg_static.s evg_kernels.hip 2>/dev/null
python3 - <<'P'
import re
lines = open('/tmp/evg_static.s').read().split('\n')
for tag, name in (("persistent f32", "_ZN3evg15evg_step_kernelIfLi64ELb1ELb0EEEvNS_8StepArgsE"), ("single-turn f32", "_ZN3evg15evg_step_kernelIfLi64ELb0ELb0EEEvNS_8StepArgsE")):
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ':'))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.end_amdhsa_kernel') or lines[i].strip().startswith('.section'))
    c = dict(valu=0, salu=0, lds=0, vmem=0, waitcnt=0)
    for l in lines[start:end]:
        t = l.strip()
        if t.startswith('v_'): c['valu'] += 1
        elif t.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif t.startswith('s_'): c['salu'] += 1
        elif t.startswith('ds_'): c['lds'] += 1
        elif t.startswith('global_') or t.startswith('buffer_') or t.startswith('flat_'): c['vmem'] += 1
    print(tag, c)
P
