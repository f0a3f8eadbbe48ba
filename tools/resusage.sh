#!/bin/bash
# Compile one kernel file with the shipped flags, keep the ISA in csrc/build/, print registers / scratch per kernel.
# usage: tools/resusage.sh fa_mfma_kernel [extra -D flags]
root=$(cd "$(dirname "$0")/.." && pwd)
f=$1; shift
mkdir -p $root/flash_attention_metal_amd/csrc/build && cd $root/flash_attention_metal_amd/csrc/build || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-division-by-zero -fno-honor-nans -fno-slp-vectorize "$@" \
  -save-temps -Rpass-analysis=kernel-resource-usage -c ../$f.hip -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy" | sed -E 's/^remark: [^ ]+ +//' | paste - - - - - |
  sed -E 's/Function Name: //; s/\[-Rpass-analysis=kernel-resource-usage\]//g' | while read -r name rest; do echo "$(echo $name | c++filt | sed 's/(fa::Params)//') | $rest"; done
