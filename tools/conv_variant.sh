#!/bin/bash
# tools/conv_variant.sh NAME 'python-expr transforming source string s' [FILE.hip] -> exp_build/lib_NAME.so
# A timing variant is a patched COPY of csrc/conv3d.hip (or FILE.hip) built with -DGFE_DIAG (the product source carries no new switch).
set -e
cd "$(dirname "$0")/../gfe-mamba_amd/csrc"
mkdir -p ../../exp_build/obj_$1
F=${3:-conv3d.hip}
python3 - "$1" "$2" "$F" <<'PY'
import sys
name, expr, fn = sys.argv[1], sys.argv[2], sys.argv[3]
s = open(fn).read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (a, s.count(a))
    s = s.replace(a, b)
exec(expr)
open("../../exp_build/obj_%s/var.hip" % name, "w").write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-pass-failed -DGFE_DIAG -I. -I../../include -c ../../exp_build/obj_$1/var.hip -o ../../exp_build/obj_$1/exp.o
OBJS=$(ls build/*.o | grep -v "build/$(basename $F .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../exp_build/obj_$1/exp.o -o ../../exp_build/lib_$1.so
echo built lib_$1.so
