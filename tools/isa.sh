#!/bin/bash
# Developer helper: rebuild the library and dump the ISA summary of one kernel (memory ops, waits, registers).
# usage: tools/isa.sh <mangled-name-substring> [file.hip]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${2:-dig_nb.hip}
make -C "$ROOT/digdriver_amd/csrc" 2>&1 | grep -E "error|Error|warning" || true
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -disable-machine-licm -I"$ROOT/include" \
    --cuda-device-only -S "$ROOT/digdriver_amd/csrc/$SRC" -o /tmp/isa/out.s 2>&1 | grep -v hip-link || true
NAME=$(grep -o "^_Z[A-Za-z0-9_]*$1[A-Za-z0-9_]*:" /tmp/isa/out.s | head -1 | tr -d ':')
echo "kernel: $NAME"
awk "/^$NAME:/,/s_endpgm/" /tmp/isa/out.s > /tmp/isa/kernel.s
grep -n "global_\|s_waitcnt\|Loop Header\|scratch_\|ds_\|buffer_" /tmp/isa/kernel.s || true
grep -A14 "\.name: *$NAME" /tmp/isa/out.s | grep -E "vgpr_count|sgpr_count|spill|lds_size" || true
wc -l /tmp/isa/kernel.s
