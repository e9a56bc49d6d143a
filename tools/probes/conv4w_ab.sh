L=$GRAFT_REPO_ROOT/exp_build/lib_w4.so
python - <<'PY'
import os, sys
sys.path.insert(0, "gfe-mamba_amd")
os.environ["GFE_HIP_LIB"] = os.environ.get("GRAFT_REPO_ROOT", ".") + "/exp_build/lib_w4.so"
import torch
from gfe_hip import nn_ops as K
import torch.nn.functional as F
BF = torch.bfloat16
ok = True
for cin, cout, shape in [(64, 64, (2, 16, 16, 24)), (128, 128, (1, 8, 16, 8)), (64, 64, (1, 11, 9, 13)), (256, 256, (1, 8, 8, 8))]:
    g = torch.Generator().manual_seed(cin + shape[1])
    B, D, H, W = shape
    x = torch.randn(B, D, H, W, cin, generator=g).to(BF).cuda()
    res = torch.randn(B, D, H, W, cout, generator=g).to(BF).cuda()
    w32 = K.pack_conv3((torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).cuda(), torch.float32)
    scale, shift = (torch.rand(B, cin, generator=g) + 0.5).cuda(), torch.randn(B, cin, generator=g).cuda()
    wb, tab = K.fold_groupnorm(w32, scale, shift, K.CONV3_TAPS, cin, cout)
    outs = {}
    for mode in ("0", "1"):
        os.environ["GFE_CONV_4W"] = mode
        y_plain = K.conv_igemm(x, wb, K.CONV3_TAPS, cout, bias_tab=tab, res=res, relu=True)
        y_stats = K.conv_igemm(x, wb, K.CONV3_TAPS, cout, bias_tab=tab, relu=True, stats=True)
        outs[mode] = (y_plain, y_stats, y_stats.gn_partials)
    eq = [bool(torch.equal(a, b)) for a, b in zip(outs["0"], outs["1"])]
    print(cin, cout, shape, "equal:", eq)
    ok = ok and all(eq)
print("ALL EQUAL" if ok else "MISMATCH")
PY
for i in 1 2; do for m in 0 1; do echo "4W=$m: $(GFE_HIP_LIB=$L GFE_CONV_4W=$m timeout 120 python tools/conv_bench.py 64 96 8 20 2>/dev/null | tail -1) | $(GFE_HIP_LIB=$L GFE_CONV_4W=$m timeout 120 python tools/conv_bench.py 64 96 8 20 64 1 2>/dev/null | tail -1)"; done; done
echo "product: $(timeout 120 python tools/conv_bench.py 64 96 8 20 2>/dev/null | tail -1)"
