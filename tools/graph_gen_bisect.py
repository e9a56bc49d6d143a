#!/usr/bin/env python3
"""Which part of the frozen generator does not reproduce itself under HIP-graph replay?  Each piece is captured on its own with static,
eagerly computed inputs and replayed three times; outputs are compared bit for bit with the eager result.   python tools/graph_gen_bisect.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfe-mamba_amd"))
import torch
from gfe_hip import det_init as det, nn_ops as K
from gfe_hip.step import build_models

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
gen, head, ft = build_models()
x = det.det_inputs(B, (96, 96, 96), seed=1)[0].cuda()


def outs(t):
    return [u for u in (t if isinstance(t, (tuple, list)) else [t]) if torch.is_tensor(u)]


def check(name, fn):
    with torch.no_grad():
        ref = [o.clone() for o in outs(fn())]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            res = outs(fn())
        rep = []
        for i in range(3):
            g.replay()
            torch.cuda.synchronize()
            rep.append([int((a != b).sum()) for a, b in zip(res, ref)])
        eag = [int((a != b).sum()) for a, b in zip(outs(fn()), ref)]
    print("%-28s eager-vs-eager mismatches %s   replay-vs-eager %s" % (name, eag, rep), flush=True)


with torch.no_grad():
    e0 = gen.encoders[0](x)
    e1 = gen.encoders[1](e0)
    e2 = gen.encoders[2](e1)
    d, h, w = e2.shape[1:4]
    mi = K.fold_mid(e2, md1=8)
    mo = gen.mid(mi)
    xm = K.fold_mid(mo, md1=8, inverse=True, shape=(d, h, w))
    d0 = gen.decoders[0](e1, xm)
e0s, e1s, e2s, mis, mos, xms, d0s = [t.clone() for t in (e0, e1, e2, mi, mo, xm, d0)]
for t, s in ((e0, e0s), (e1, e1s)):
    if getattr(t, "gn_partials", None) is not None:
        s.gn_partials = t.gn_partials.clone()
check("encoders[0] (first block)", lambda: gen.encoders[0](x))
check("encoders[1]", lambda: gen.encoders[1](e0s))
check("encoders[2]", lambda: gen.encoders[2](e1s))
check("fold_mid + ViT + unfold", lambda: K.fold_mid(gen.mid(K.fold_mid(e2s, md1=8)), md1=8, inverse=True, shape=(d, h, w)))
check("decoders[0]", lambda: gen.decoders[0](e1s, xms))
check("decoders[1]", lambda: gen.decoders[1](e0s, d0s))
check("whole generator", lambda: gen(x, output_vit_mid=True))

# ---- inside encoders[1] ----------------------------------------------------------------------------------------------------------
if os.environ.get("BISECT_DEEP", "1") == "1":
    import torch.nn.functional as F
    blk = gen.encoders[1].basic_module
    with torch.no_grad():
        p0 = K.maxpool2(e0s).clone()
        r0 = blk.lift(p0)
        r0s = r0.clone(); r0s.gn_partials = r0.gn_partials.clone()
        o0 = blk._conv2_through_lift(p0, r0s)
        o0s = o0.clone(); o0s.gn_partials = o0.gn_partials.clone()
    check("e1: maxpool", lambda: K.maxpool2(e0s))
    check("e1: lift (1x1 conv + stats)", lambda: (lambda r: (r, r.gn_partials))(blk.lift(p0)))
    check("e1: conv2 through lift", lambda: (lambda o: (o, o.gn_partials))(blk._conv2_through_lift(p0, r0s)))
    check("e1: conv3 + residual", lambda: blk.conv3(o0s, residual=r0s))

    def c2_parts():
        c1, sc = blk.conv1, blk.conv2
        gn, conv = sc.groupnorm, sc.conv
        cin, c, cout = c1.in_channels, c1.out_channels, conv.out_channels
        w32, w2m, w1, b1, g, b = blk._pack2.get([conv.weight, c1.weight, c1.bias, gn.weight, gn.bias], lambda: None)
        scale, shift = K.groupnorm_scale_shift(r0s, g, b, gn.num_groups, gn.eps)
        Bn, cp, nslab = scale.shape[0], w32.shape[2], (cin + 31) // 32
        rhs = (scale.t().unsqueeze(2) * w1.unsqueeze(1)).reshape(c, Bn * cin)
        weff = K.gemm_f32(w2m, False, rhs, True).view(27 * cp, Bn, cin)
        weff16 = weff.view(27, cp, Bn, nslab, 32).permute(2, 3, 0, 1, 4).contiguous().to(K.BF16)
        _, tab = K.fold_groupnorm(w32, scale, scale * b1 + shift, K.CONV3_TAPS, c, cout)
        return scale, shift, weff, weff16, tab
    check("e1: conv2 parts (scale, shift, weff f32, weff bf16, tab)", c2_parts)
    with torch.no_grad():
        sc_, sh_, weff_, weff16_, tab_ = [t.clone() for t in c2_parts()]
    check("e1: conv_igemm(x, weff, tab) no stats", lambda: K.conv_igemm(p0, weff16_, K.CONV3_TAPS, 128, bias_tab=tab_, relu=True))
    check("e1: conv_igemm(x, weff, tab) + stats", lambda: (lambda o: (o, o.gn_partials))(K.conv_igemm(p0, weff16_, K.CONV3_TAPS, 128, bias_tab=tab_, relu=True, stats=True)))
    sc3 = blk.conv3
    g3, b3 = sc3.groupnorm.weight.detach().float().contiguous(), sc3.groupnorm.bias.detach().float().contiguous()
    w32_3 = K.pack_conv3(sc3.conv.weight, torch.float32)
    check("e1: groupnorm_scale_shift from partials", lambda: K.groupnorm_scale_shift(o0s, g3, b3, 8, 1e-5))
    with torch.no_grad():
        s3, t3 = [t.clone() for t in K.groupnorm_scale_shift(o0s, g3, b3, 8, 1e-5)]
    check("e1: fold_groupnorm (w, tab)", lambda: K.fold_groupnorm(w32_3, s3, t3, K.CONV3_TAPS, 128, 128))
    with torch.no_grad():
        wb3, tab3 = [t.clone() for t in K.fold_groupnorm(w32_3, s3, t3, K.CONV3_TAPS, 128, 128)]
    check("e1: conv_igemm 128->128 + res", lambda: K.conv_igemm(o0s, wb3, K.CONV3_TAPS, 128, bias_tab=tab3, res=r0s, relu=True))
    check("e1: conv_igemm 128->128 + res + stats", lambda: (lambda o: (o, o.gn_partials))(K.conv_igemm(o0s, wb3, K.CONV3_TAPS, 128, bias_tab=tab3, res=r0s, relu=True, stats=True)))
