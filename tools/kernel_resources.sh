#!/bin/bash
# usage: tools/kernel_resources.sh pk_prog_cgrid_fast.hip [extra hipcc flags]  -- registers / scratch / occupancy of every kernel of one TU
cd "$(dirname "$0")/../parcels_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function --offload-device-only \
  -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /tmp/kr_$$.o 2>&1 | grep remark | sed 's/.*remark: //; s/ \[-Rpass.*//' | \
  awk '/Function Name/{if(l)print l; l=$0; next} {gsub(/^ +/,""); l=l" | "$0} END{print l}' | \
  sed 's/Function Name: //; s/ | AGPRs: 0//; s/ | Dynamic Stack: False//; s/ | LDS Size \[bytes\/block\]: 0//' | c++filt
rm -f /tmp/kr_$$.o
