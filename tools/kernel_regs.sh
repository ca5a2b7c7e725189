#!/bin/bash
# register / scratch / LDS budget of the library's kernels (device-only assembly of one translation unit):
#   tools/kernel_regs.sh [rnde.hip] [pattern] [-DFLAGS...]
cd "$(dirname "$0")/.."
SRC=${1:-rnde.hip}; PAT=${2:-attempt}; shift 2 2>/dev/null
OUT=/tmp/${SRC%.hip}$(echo "$@" | tr -c 'A-Za-z0-9' '_').s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-undefined-internal -Wno-pass-failed "$@" --cuda-device-only -S regneuralde.jl_amd/csrc/$SRC -o $OUT || exit 1
python3 - "$OUT" "$PAT" <<'P'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    name, body = m.group(1), m.group(2)
    if sys.argv[2] not in name:
        continue
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", body) or [0, "?"])[1]
    print(f"{name[:110]:110s} vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>4} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}")
P
echo "asm: $OUT"
