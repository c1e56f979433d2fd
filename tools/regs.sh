#!/bin/bash
# dev aid: VGPR / spill / LDS figures of every kernel in a compiled object's gfx950 code object
# usage: tools/regs.sh guided-vae-nmf_amd/csrc/engine.hip [filter]
set -e
D=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 --cuda-device-only -S -o $D/k.s "$1" ${EXTRA}
python3 - "$D/k.s" "${2:-}" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2]
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if flt and flt not in name: continue
    g = lambda k: (re.search(r"\.amdhsa_%s (\S+)" % k, body) or [None, "?"])[1]
    print(name[:110], "vgpr", g("next_free_vgpr"), "accum_off", g("accum_offset"), "scratch", g("private_segment_fixed_size"))
PY
rm -rf $D
