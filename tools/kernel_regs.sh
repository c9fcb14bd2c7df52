#!/bin/bash
# usage: tools/kernel_regs.sh <object.o> [name regex]  -> VGPR / AGPR / SGPR / scratch bytes / LDS of every kernel in the gfx950 code object
set -e
obj=$1; filt=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $obj
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.o > $tmp/notes.txt
python3 - "$tmp/notes.txt" "$filt" <<'PY'
import sys, re
txt = open(sys.argv[1]).read()
for blk in txt.split('- .agpr_count:')[1:]:
    blk = '.agpr_count:' + blk
    g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    if re.search(sys.argv[2], name):
        print(f"{name[:120]:120s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>7s}")
PY
rm -rf $tmp
