#!/bin/bash
# Registers / spills / occupancy of every kernel of one source: tools/kernel_resources.sh se3et_amd/csrc/kpconv_mfma.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/_kr.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "remark:" | sed -E 's/^.*remark: [^ ]+ +//; s/ \[-Rpass.*$//' \
  | sed -E 's/^.*remark: +//' | awk '/Function Name|^Name:/ {if (line) print line; line=$0; next} /VGPRs:|AGPRs:|Spill|Occupancy|LDS Size|ScratchSize/ {line=line " | " $0} END {print line}'
