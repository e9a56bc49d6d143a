#!/bin/bash
# Timing-experiment builds of libgfe_hip.so: tools/build_exp.sh NAME "-DGFE_EXP_X ..." -> exp_build/lib_NAME.so (not shipped; GFE_HIP_LIB selects it)
set -e
cd "$(dirname "$0")/../gfe-mamba_amd/csrc"
mkdir -p ../../exp_build/obj_$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-pass-failed -DGFE_DIAG $2 -c ${3:-conv3d.hip} -o ../../exp_build/obj_$1/exp.o
OBJS=$(ls build/*.o | grep -v "build/$(basename ${3:-conv3d.hip} .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../exp_build/obj_$1/exp.o -o ../../exp_build/lib_$1.so
