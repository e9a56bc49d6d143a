#!/bin/bash
# Rebuilds the round-6 attention experiment (two software-pipelined 32-row sub-tiles per wave) as a DIAGNOSTIC library: exp_build/lib_attn2.so.
# The product tree is not touched: the sources are copied into a scratch directory, attn.hip gets the dispatch patch there.
set -e
R="$(cd "$(dirname "$0")/../../.." && pwd)"
S=$R/exp_build/attn2_src; rm -rf $S; mkdir -p $S $R/exp_build
cp $R/gfe-mamba_amd/csrc/*.hip $R/gfe-mamba_amd/csrc/*.h $S/
cp $R/tools/probes/attn2_experiment/attn2.hip.txt $S/attn2.hip
cp $R/tools/probes/attn2_experiment/attn_params.h.txt $S/attn_params.h
(cd $S && patch -p3 attn.hip < <(sed -n '/^diff --git a\/gfe-mamba_amd\/csrc\/attn.hip/,/^diff --git a\/gfe-mamba_amd\/csrc\/Makefile/p' $R/tools/probes/attn2_experiment/product.patch | sed '$d'))
sed -i 's#"../../include/gfe_hip.h"#"'$R'/include/gfe_hip.h"#' $S/common.h
cd $S
for f in *.hip; do
  X=""; case $f in attn.hip|attn2.hip|sscan2.hip) X="-fno-honor-nans";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-pass-failed -DGFE_DIAG $X -I. -c $f -o ${f%.hip}.o &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o $R/exp_build/lib_attn2.so
echo built exp_build/lib_attn2.so
