#!/bin/bash
# device assembly of one kernel of la3dm_hip.hip: tools/check/kasm.sh <mangled-name substring> [out.s]
# prints register / scratch / LDS use and writes the kernel's text to out.s (default /tmp/asm/k.s)
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
mkdir -p /tmp/asm
( cd "$ROOT/la3dm_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -S --cuda-device-only ${SRC:-la3dm_hip.hip} -o /tmp/asm/all.s 2>/tmp/asm/err.txt ) || { grep -A5 error /tmp/asm/err.txt; exit 1; }
name=$(grep -o "^_Z[A-Za-z0-9_]*$1[A-Za-z0-9_]*:" /tmp/asm/all.s | head -1 | tr -d ':')
[ -n "$name" ] || { echo "no kernel matches $1"; exit 1; }
out=${2:-/tmp/asm/k.s}
awk -v n="$name:" 'index($0,n)==1{p=1} p{print} /\.Lfunc_end/{if(p) exit}' /tmp/asm/all.s > "$out"
echo "$name -> $out ($(grep -c '^\s*v_' "$out") VALU lines)"
grep "$name\.\(num_vgpr\|num_agpr\|numbered_sgpr\|private_seg_size\)" /tmp/asm/all.s | sed 's/.*\.\(num_vgpr\|num_agpr\|numbered_sgpr\|private_seg_size\)/  \1/'
grep -A40 "amdhsa_kernel $name" /tmp/asm/all.s | grep "group_segment_fixed_size" 
