#!/bin/bash
# development: device assembly of mrp_engine_kernels.hip (-> /tmp/xe.s) and ONE kernel of it alone (-> /tmp/xe_kernel.s); usage: xe_isa.sh [kernel name fragment]
K=${1:-mrp_cross_emit_kernel}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o /tmp/xe.s mrp_engine_kernels.hip 2>&1 | grep -v "warning\|^$"
awk -v k="$K" 'index($0, "_Z") == 1 && index($0, k) > 0 && index($0, ": ") > 0 {on=1} on {print} on && /\.end_amdhsa_kernel/ {exit}' /tmp/xe.s > /tmp/xe_kernel.s
grep "next_free_vgpr\|next_free_sgpr\|group_segment_fixed_size\|private_segment_fixed" /tmp/xe_kernel.s; echo "scratch ops: $(grep -c scratch_ /tmp/xe_kernel.s), lines $(wc -l < /tmp/xe_kernel.s)"
