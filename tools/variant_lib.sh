#!/bin/bash
# builds a VARIANT of the HIP library next to the product's (slslam_amd/_lib/variants/<name>.so, travels with the gpurun snapshot) with extra
# compiler flags; run a tool against it with SLSLAM_HIP_LIBRARY=<path>.   tools/variant_lib.sh <name> <flags...>
cd /root/repo; NAME=$1; shift
mkdir -p slslam_amd/_lib/variants
C=slslam_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics "$@" $C/lba_api.hip $C/lba_pack.cpp $C/po_api.hip $C/ransac_api.hip -o slslam_amd/_lib/variants/$NAME.so 2>&1 | grep -E "error|scratch" | head
ls -la slslam_amd/_lib/variants/$NAME.so
