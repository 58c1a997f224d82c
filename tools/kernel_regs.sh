#!/bin/bash
# Register / scratch / LDS use of every kernel in a .hip file (gfx950 ISA from hipcc --save-temps).  usage: tools/kernel_regs.sh rt_kernels.hip [extra flags]
set -e
SRC=$(realpath "$1"); shift
TMP=$(mktemp -d /tmp/kregs.XXXX)
cd "$TMP"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC "$@" --save-temps -c "$SRC" -o k.o 2>/dev/null
S=$(ls *gfx950*.s)
awk '/\.amdhsa_kernel /{k=$2} /\.amdhsa_next_free_vgpr/{v=$2} /\.amdhsa_next_free_sgpr/{s=$2} /\.amdhsa_group_segment_fixed_size/{l=$2} /\.amdhsa_private_segment_fixed_size/{p=$2} /\.end_amdhsa_kernel/{printf "%-90s vgpr %4s sgpr %4s lds %6s scratch %5s\n", k, v, s, l, p}' "$S"
echo "ISA: $TMP/$S"
