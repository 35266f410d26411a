#!/bin/bash
# Compact per-kernel register / scratch / occupancy table for one HIP source:  tools/regs.sh csrc-file [filter]
f=$1; pat=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I sglang_security_amd/csrc -c "$f" -o /tmp/regs_tmp.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy" \
  | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - | grep -E "$pat" \
  | sed -E 's/Function Name: //; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/'
