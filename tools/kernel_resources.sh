#!/bin/bash
# Registers, scratch and occupancy of every kernel of csrc/ as the compiler reports them (-Rpass-analysis=kernel-resource-usage), built with the
# flags of csrc/Makefile (per-object flags included): one line per kernel.    usage: tools/kernel_resources.sh > profiles/rNN_kernel_resource_usage.txt
# (no GPU needed: hipcc cross-compiles gfx950)
cd "$(dirname "$0")/../posegraph-ceres_amd/csrc" || exit 1
for f in *.hip; do
  flags=$(make -n -B "obj/${f%.hip}.o" 2>/dev/null | grep -m1 hipcc | sed 's/^.*hipcc //; s/ -x hip .*//')
  /opt/rocm/bin/hipcc $flags -x hip --cuda-device-only -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|    VGPRs:|ScratchSize|Occupancy|VGPRs Spill|LDS Size" | sed 's/.*remark: *//; s/ *\[-Rpass.*//; s/^ *//' |
    paste - - - - - - | sed "s/^/$f: /" |
    if [ "$f" = pgo_res_kernels.hip ]; then grep "k_res_"; else cat; fi     # (that unit is pgo_kernels.hip's text again; only the resident stream's kernels are launched from it)
done
