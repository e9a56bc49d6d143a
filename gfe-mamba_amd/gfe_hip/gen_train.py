"""Differentiable (training-mode) forward of the generator -- SURVEY 8-f1: conv3d backward and the generator training step of
main_gan_vit.py:68-82 (`images = model(condition); loss = L1(images, real) ...; backward; optimizer.step`).

The inference path (pytorch3dunet/unet3d/*.py) folds GroupNorm into per-sample conv weights and never stores what a backward would need.
Here every layer is a torch.autograd.Function over the same HIP kernels:

  conv 3x3x3        forward  gfe_conv3d (implicit GEMM) on the materialised GroupNorm output x^ = s x + t (gfe_gn_apply)
                    dgrad    the SAME kernel with flipped taps and transposed weights
                    wgrad    voxel-reduction GEMMs on the bf16 MFMA GEMM (gfe_gemm_ex, both operands reduction-major): one per tap over
                             spatially zero-padded copies, so a tap is a row offset (no boundary logic, no gather)
  GroupNorm         forward  gfe_groupnorm_scale_shift + gfe_gn_apply;  backward gfe_gn_bwd_sums + gfe_gn_bwd_apply (two HBM passes)
  ReLU / residual   backward gfe_mask_relu_bf16 from the stored output; the sum's gradient fans out
  MaxPool3d(2)      gfe_maxpool2 / gfe_maxpool2_bwd
  ConvTranspose3d(k3, s2, p1) + nearest resize + skip sum   forward: the fused inference kernel; backward: the resize's duplicate plane is
                    summed back, the 8 parity classes of the upsampled gradient are stacked along channels and ONE 8-tap conv gives dx (f32
                    accumulation over all classes); the weight gradient is the same voxel-reduction GEMM on that stacked tensor
  1x1x1 convs       conv_in1 / conv_igemm (1 tap) / conv_out1 forward; gfe_conv_in1_wgrad, gfe_gemm_ex, gfe_conv_out1_bwd backward
  bottleneck ViT    patchify / un-patchify as index views, LayerNorm (gfe_layernorm_rows), Linears (train_ops.linear: bf16 MFMA for the
                    147k-wide patch embedding, exact f32 below 4096 inputs), 25-token attention (gfe_sdpa_small), exact-erf GELU

Activations are channels-last bf16; parameter gradients are f32.  Dropout follows module.training (main_gan_vit trains with it on; the
parity tests run eval mode so that the reference's autograd gradients are deterministic)."""
import os

import torch
import torch.nn.functional as F

from . import call, lib, nn_ops as K, ptr, stream
from .head_ops import dropout, gelu, layernorm_rows, sdpa_small
from .nn_ops import BF16
from .train_ops import linear

# Tests only: when a list, every ReLU mask and max-pool selection of the forward is appended in call order (channels-last bool tensors), so
# that a reference autograd can be evaluated at the same activation pattern (oracle.ref_ops.activation_pattern).
PATTERN_LOG = None


# ---- thin wrappers of csrc/gen_train.hip -----------------------------------------------------------------------------------------
def gn_apply(x, scale, shift):
    B, C = x.shape[0], x.shape[-1]
    y = torch.empty_like(x)
    call("gfe_gn_apply", ptr(x), ptr(scale), ptr(shift), ptr(y), B, x.numel() // (B * C), C, stream())
    return y


def mask_relu(dy, y):
    out = torch.empty_like(dy)
    call("gfe_mask_relu_bf16", ptr(dy), ptr(y), ptr(out), dy.numel(), stream())
    return out


def conv_wgrad(inp, dout, taps):
    """dW[tap][co][ci] = sum_voxels dout[v][co] * inp[v + tap][ci] (zero outside the volume) for channels-last bf16 (B, D, H, W, C) tensors:
    one reduction-major GEMM per tap over spatially zero-padded copies (a tap is then a row offset).  Returns (ntaps, Co, Ci) f32."""
    B, D, H, W, Ci = inp.shape
    Co = dout.shape[-1]
    if Ci % 32 == 0 and Co % 64 == 0 and len(taps) <= 27 and os.environ.get("GFE_WGRAD_GEMM") != "1":
        # one fused launch for all taps (csrc/conv_wgrad.hip): each tensor is read about once instead of once per tap
        ns = lib().gfe_conv3d_wgrad_splits(B, D, H, W, Ci, Co)
        assert ns > 0, ns
        part = torch.empty((ns, len(taps), Co, Ci), dtype=torch.float32, device=inp.device)
        out = torch.empty((len(taps), Co, Ci), dtype=torch.float32, device=inp.device)
        tarr, tptr = K._i8(taps)
        call("gfe_conv3d_wgrad", ptr(inp.contiguous()), ptr(dout.contiguous()), ptr(part), ptr(out), tptr, len(taps), B, D, H, W, Ci, Co, stream())
        return out
    xp = F.pad(inp, (0, 0, 1, 1, 1, 1, 1, 1)).reshape(-1, Ci)
    dp = F.pad(dout, (0, 0, 1, 1, 1, 1, 1, 1)).reshape(-1, Co)
    R = xp.shape[0]
    out = torch.zeros((len(taps), Co, Ci), dtype=torch.float32, device=inp.device)
    for i, (od, oh, ow) in enumerate(taps):
        o = (od * (H + 2) + oh) * (W + 2) + ow
        r0, r1 = max(0, -o), R - max(0, o)
        K.gemm_ex(dp[r0:r1], True, xp[r0 + o:r1 + o], True, accum_into=out[i], split_k=max(1, min(256, (r1 - r0) // 4096)))
    return out


# ---- layers ----------------------------------------------------------------------------------------------------------------------
class _GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps):
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        xd = x.detach()
        part = getattr(x, "gn_partials", None)             # written by the producing conv's epilogue: no statistics pass over x
        if part is not None:
            xd.gn_partials = part
        scale, shift = K.groupnorm_scale_shift(xd, g, b, groups, eps)                 # (B, C): x^ = scale * x + shift
        xhat = gn_apply(xd, scale, shift)
        rstd = scale / g                                                              # scale = gamma * rstd, shift = beta - mu * scale
        mu = (b - shift) / scale
        ctx.save_for_backward(x.detach(), g, mu.contiguous(), rstd.contiguous())
        ctx.groups = groups
        return xhat

    @staticmethod
    def backward(ctx, dxhat):
        x, g, mu, rstd = ctx.saved_tensors
        B, C = x.shape[0], x.shape[-1]
        V = x.numel() // (B * C)
        G = ctx.groups
        d = dxhat.contiguous()
        S = torch.zeros((2, B, C), dtype=torch.float32, device=x.device)
        ws = torch.empty(B * int(lib().gfe_gn_bwd_sums_blocks(V)) * 2 * C, dtype=torch.float32, device=x.device)    # per-block partial rows, folded in order
        call("gfe_gn_bwd_sums", ptr(d), ptr(x), ptr(mu), ptr(rstd), ptr(S[0]), ptr(S[1]), ptr(ws), B, V, C, stream())
        m = float((C // G) * V)
        ca = ((g * S[0]).view(B, G, C // G).sum(-1, keepdim=True) / m).expand(B, G, C // G).reshape(B, C).contiguous()
        cb = ((g * S[1]).view(B, G, C // G).sum(-1, keepdim=True) / m).expand(B, G, C // G).reshape(B, C).contiguous()
        dx = torch.empty_like(x)
        call("gfe_gn_bwd_apply", ptr(d), ptr(x), ptr(mu), ptr(rstd), ptr(g), ptr(ca), ptr(cb), None, ptr(dx), B, V, C, stream())
        return dx, S[1].sum(0), S[0].sum(0), None, None


class _Conv3Fn(torch.autograd.Function):
    """y = [relu](conv3x3x3(x^, W) [+ res]) on channels-last bf16 (buildingblocks.py:46-52: padding 1, no bias)."""

    @staticmethod
    def forward(ctx, xhat, weight, res, relu):
        cout = weight.shape[0]
        y = K.conv_igemm(xhat.detach(), K.pack_conv3(weight), K.CONV3_TAPS, cout, res=None if res is None else res.detach(), relu=relu, stats=True)
        if relu and PATTERN_LOG is not None:
            PATTERN_LOG.append(y > 0)
        ctx.save_for_backward(xhat.detach(), weight, y if relu else None)
        ctx.has_res, ctx.relu = res is not None, relu
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, weight, y = ctx.saved_tensors
        d = dy.contiguous()
        if ctx.relu:
            d = mask_relu(d, y)
        cin = weight.shape[1]
        dx = dw = None
        if ctx.needs_input_grad[0]:                   # dgrad: cross-correlation with the taps flipped and in / out channels swapped
            dx = K.conv_igemm(d, K.pack_conv3(weight.detach().transpose(0, 1).flip(2, 3, 4)), K.CONV3_TAPS, cin)
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad(xhat, d, K.CONV3_TAPS).permute(1, 2, 0).reshape(weight.shape).to(weight.dtype)
        return dx, dw, (d if ctx.has_res else None), None


class _Conv1Fn(torch.autograd.Function):
    """1x1x1 conv with bias, Cin >= 8 (ResNetBlock.conv1 of the deeper encoders, buildingblocks.py:191-198)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        cout, cin = weight.shape[:2]
        if K.conv1x1_ok(cin, cout):                   # the streaming product (conv1x1.hip); other widths stay on the one-tap implicit GEMM
            y = K.conv1x1(x.detach(), weight.detach().reshape(cout, cin).to(torch.bfloat16).contiguous(), bias.detach().float().contiguous(), stats=True)
        else:
            y = K.conv_igemm(x.detach(), K.pack_conv1(weight), [(0, 0, 0)], cout, bias=bias.detach().float().contiguous(), stats=True)
        ctx.save_for_backward(x.detach(), weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        d = dy.contiguous()
        cout, cin = weight.shape[:2]
        dx = dw = None
        if ctx.needs_input_grad[0]:                   # dx = dy W: the same product with the channels swapped
            if K.conv1x1_ok(cout, cin):
                dx = K.conv1x1(d, weight.detach().reshape(cout, cin).t().to(torch.bfloat16).contiguous(), None, stats=False)
            else:
                dx = K.conv_igemm(d, K.pack_conv1(weight.detach().transpose(0, 1).contiguous()), [(0, 0, 0)], cin)
        d2, x2 = d.view(-1, cout), x.view(-1, cin)
        if ctx.needs_input_grad[1]:
            dw = K.gemm_ex(d2, True, x2, True, split_k=max(1, min(256, d2.shape[0] // 4096))).view(weight.shape).to(weight.dtype)
        db = torch.zeros(cout, dtype=torch.float32, device=d.device)
        ws = torch.empty(int(lib().gfe_gen_rows_blocks(d2.shape[0])) * 2 * cout, dtype=torch.float32, device=d.device)
        call("gfe_conv_in1_wgrad", None, ptr(d2), None, ptr(db), ptr(ws), d2.shape[0], cout, stream())
        return dx, dw, db


class _LiftIn1Fn(torch.autograd.Function):
    """First ResNetBlock.conv1: one-channel f32 volume (B, 1, D, H, W) -> (B, D, H, W, C) bf16, r_c = w_c x + b_c."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        r = K.conv_in1(x.detach(), weight.detach().float().view(-1).contiguous(), bias.detach().float().contiguous())
        ctx.save_for_backward(x.detach())
        ctx.wshape = weight.shape
        return r

    @staticmethod
    def backward(ctx, dr):
        (x,) = ctx.saved_tensors
        C = dr.shape[-1]
        d = dr.contiguous().view(-1, C)
        g = torch.zeros((2, C), dtype=torch.float32, device=d.device)
        ws = torch.empty(int(lib().gfe_gen_rows_blocks(d.shape[0])) * 2 * C, dtype=torch.float32, device=d.device)
        call("gfe_conv_in1_wgrad", ptr(x.contiguous().float()), ptr(d), ptr(g[0]), ptr(g[1]), ptr(ws), d.shape[0], C, stream())
        return None, g[0].view(ctx.wshape), g[1]


def _maxpool_route(x, dy):
    """dx of nn.MaxPool3d(2): every dy goes to the first maximum of its window (ATen's tie rule)."""
    B, D, H, W, C = x.shape
    dx = torch.zeros_like(x)
    call("gfe_maxpool2_bwd", ptr(x), ptr(dy.contiguous()), ptr(dx), B, D, H, W, C, stream())
    return dx


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.detach()
        ctx.save_for_backward(x)
        y = K.maxpool2(x)
        if PATTERN_LOG is not None:
            PATTERN_LOG.append(_maxpool_route(x, torch.ones_like(y)) > 0)
        return y

    @staticmethod
    def backward(ctx, dy):
        return _maxpool_route(ctx.saved_tensors[0], dy)


_CT_TAPS8 = [(od, oh, ow) for od in (-1, 0) for oh in (-1, 0) for ow in (-1, 0)]


def _ct_axis(parity, offset):
    """Kernel index k of ConvTranspose3d(k3, s2, p1) that links input i to upsampled index 2 (i + offset) + parity (= 2i + k - 1), or None."""
    if parity == 0:
        return 1 if offset == 0 else None
    return 0 if offset == -1 else 2


class _UpJoinFn(torch.autograd.Function):
    """encoder_features + nearest_resize(ConvTranspose3d(k3, s2, p1)(x)) (buildingblocks.py:396-400, 523-537)."""

    @staticmethod
    def forward(ctx, enc, x, weight, module):
        out = module(enc.detach(), x.detach())                      # the fused inference kernel (no GroupNorm inside)
        ctx.save_for_backward(x.detach(), weight)
        ctx.oshape = enc.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, D, H, W, cin = x.shape
        cout = weight.shape[1]
        g = dy.contiguous()
        # nearest resize 2n-1 -> 2n duplicates the first plane (dst j <- src max(j-1, 0)): its adjoint adds plane 0 into plane 1 and drops it
        for ax, n in ((1, D), (2, H), (3, W)):
            if g.shape[ax] == 2 * n:
                head = g.narrow(ax, 0, 1)
                g = g.narrow(ax, 1, 2 * n - 1).clone()
                g.narrow(ax, 0, 1).add_(head)
        # the 8 parity classes of the upsampled gradient, zero-padded to the input's size, stacked along channels: (B, D, H, W, 8 * Cout)
        parts = []
        for pd in (0, 1):
            for ph in (0, 1):
                for pw in (0, 1):
                    c = g[:, pd::2, ph::2, pw::2]
                    parts.append(F.pad(c, (0, 0, 0, W - c.shape[3], 0, H - c.shape[2], 0, D - c.shape[1])))
        pcat = torch.cat(parts, dim=-1).contiguous()
        dx = dw = None
        wd = weight.detach().float()                                 # (Cin, Cout, 3, 3, 3)
        if ctx.needs_input_grad[1]:
            # dx[i][ci] = sum over (class, offset) of W[ci][co][k(class, offset)] * class[i + offset][co]: one 8-tap conv over the stacked classes
            wt = torch.zeros((8, cin, 8 * cout), dtype=torch.float32, device=x.device)
            for t, (od, oh, ow) in enumerate(_CT_TAPS8):
                for ci_, (pd, ph, pw) in enumerate((a, b_, c_) for a in (0, 1) for b_ in (0, 1) for c_ in (0, 1)):
                    kd, kh, kw = _ct_axis(pd, od), _ct_axis(ph, oh), _ct_axis(pw, ow)
                    if kd is not None and kh is not None and kw is not None:
                        wt[t, :, ci_ * cout:(ci_ + 1) * cout] = wd[:, :, kd, kh, kw]
            dx = K.conv_igemm(pcat, K._pack(wt, 8 * cout), _CT_TAPS8, cin)
        if ctx.needs_input_grad[2]:
            # dW[ci][co][k] = sum_i x[i][ci] * class[i + offset][co]: the voxel-reduction GEMM with x in the role of the output gradient
            dwt = conv_wgrad(pcat, x, _CT_TAPS8)                      # (8 taps, Cin, 8 * Cout)
            dw = torch.zeros_like(wd)
            for t, (od, oh, ow) in enumerate(_CT_TAPS8):
                for ci_, (pd, ph, pw) in enumerate((a, b_, c_) for a in (0, 1) for b_ in (0, 1) for c_ in (0, 1)):
                    kd, kh, kw = _ct_axis(pd, od), _ct_axis(ph, oh), _ct_axis(pw, ow)
                    if kd is not None and kh is not None and kw is not None:
                        dw[:, :, kd, kh, kw] += dwt[t, :, ci_ * cout:(ci_ + 1) * cout]
            dw = dw.to(weight.dtype)
        return dy, dx, dw, None


_HOST_SCALARS = {}


def _host_scalar(p):
    """float(p) for a one-element parameter without a device-to-host sync per step: read once per parameter VALUE (keyed on the storage
    address, the version counter and the optimizer's update counter, which FlatAdam bumps when its kernel rewrites the storage)."""
    key = (p.data_ptr(), p._version, getattr(p, "_gfe_epoch", (0,))[0])
    hit = _HOST_SCALARS.get(id(p))
    if hit is None or hit[0] != key:
        hit = (key, float(p.detach().float().item()))
        _HOST_SCALARS[id(p)] = hit
    return hit[1]


class _Out1Fn(torch.autograd.Function):
    """final_conv: 1x1x1, C -> 1, bias; f32 (B, 1, D, H, W) output (model.py:123, 162)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        w = weight.detach().float().view(-1).contiguous()
        ctx.save_for_backward(x.detach(), w)
        ctx.wshape = weight.shape
        return K.conv_out1(x.detach(), w, _host_scalar(bias))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        C = x.shape[-1]
        d = dy.contiguous().float().view(-1)
        dx = torch.empty_like(x)
        g = torch.zeros(C + 1, dtype=torch.float32, device=x.device)
        ws = torch.empty(int(lib().gfe_gen_rows_blocks(d.numel())) * (C + 1), dtype=torch.float32, device=x.device)
        call("gfe_conv_out1_bwd", ptr(x), ptr(d), ptr(w), ptr(dx), ptr(g[:C]), ptr(g[C:]), ptr(ws), d.numel(), C, stream())
        return dx, g[:C].view(ctx.wshape), g[C:].view(1)


def single_conv(sc, x, residual=None):
    """SingleConv (order 'gcr' / 'gc'): GroupNorm -> Conv3d [-> ReLU]; with a residual the block tail relu(conv + r)."""
    gn = sc.groupnorm
    xhat = _GroupNormFn.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps)
    return _Conv3Fn.apply(xhat, sc.conv.weight, residual, sc.relu or residual is not None)


def resnet_block(blk, x):
    """ResNetBlock.forward (buildingblocks.py:218-229): r = conv1(x); relu(conv3(conv2(r)) + r)."""
    c1 = blk.conv1
    if isinstance(c1, torch.nn.Identity):
        r = x
    elif c1.in_channels == 1:
        r = _LiftIn1Fn.apply(x, c1.weight, c1.bias)
    else:
        r = _Conv1Fn.apply(x, c1.weight, c1.bias)
    o = single_conv(blk.conv2, r)
    return single_conv(blk.conv3, o, residual=r)


def _dropout(x, module):
    return dropout(x, module.p, module.training)


def vit_forward_train(vit, img):
    """vit_pytorch_diy.ViT.forward (vit.py:124-137) on a channels-last bf16 image (B, H, W, C), differentiable."""
    B, Himg, Wimg, C = img.shape
    p, n, dim = vit.patch, vit.num_patches, vit.dim
    hh, ww = Himg // p, Wimg // p
    tpe, fpe = vit.to_patch_embedding, vit.from_patch_embedding
    # 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' on channels-last memory (vit.py:96)
    t = img.view(B, hh, p, ww, p, C).permute(0, 1, 3, 2, 4, 5).reshape(B, n, p * p * C).float()
    t = layernorm_rows(t, tpe[1].weight, tpe[1].bias, tpe[1].eps)
    t = linear(t, tpe[2].weight, tpe[2].bias)
    t = layernorm_rows(t, tpe[3].weight, tpe[3].bias, tpe[3].eps)
    x = torch.cat((vit.cls_token.expand(B, -1, -1), t), dim=1)                       # vit.py:127-129
    x = x + vit.pos_embedding[:, :n + 1]
    x = _dropout(x, vit.dropout)
    for attn, ff in vit.transformer.layers:                                           # vit.py:76-79
        h = layernorm_rows(x, attn.norm.weight, attn.norm.bias, attn.norm.eps)
        q, k, v = linear(h, attn.to_qkv.weight, None).chunk(3, dim=-1)
        p_attn = attn.dropout.p if (attn.training and attn.dropout.training) else 0.0  # `attn = self.dropout(attn)` (vit.py:59): inside the kernel
        o = sdpa_small(q.contiguous(), k.contiguous(), v.contiguous(), attn.heads, causal=False, dropout_p=p_attn)
        x = _dropout(linear(o, attn.to_out[0].weight, attn.to_out[0].bias), attn.to_out[1]) + x
        h = layernorm_rows(x, ff.net[0].weight, ff.net[0].bias, ff.net[0].eps)
        h = _dropout(gelu(linear(h, ff.net[1].weight, ff.net[1].bias)), ff.net[3])
        x = _dropout(linear(h, ff.net[4].weight, ff.net[4].bias), ff.net[5]) + x
    x = layernorm_rows(x, vit.transformer.norm.weight, vit.transformer.norm.bias, vit.transformer.norm.eps)
    # from_patch_embedding (vit.py:102-110)
    x = layernorm_rows(x, fpe[0].weight, fpe[0].bias, fpe[0].eps)
    x = linear(x.transpose(1, 2).contiguous(), fpe[2].weight, fpe[2].bias).transpose(1, 2).contiguous()      # Linear over the token axis
    x = linear(x, fpe[4].weight, fpe[4].bias)
    x = layernorm_rows(x, fpe[5].weight, fpe[5].bias, fpe[5].eps)
    out = x.view(B, hh, ww, p, p, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Himg, Wimg, C)               # un-patchify (vit.py:109)
    return out.to(BF16)


def _fold(x, md1=8):
    """'b c (md1 md2) h w -> b c (h md1) (md2 w)' (model.py:150) on channels-last (B, D, H, W, C) -> (B, H*md1, md2*W, C)."""
    B, D, H, W, C = x.shape
    return x.view(B, md1, D // md1, H, W, C).permute(0, 3, 1, 2, 4, 5).reshape(B, H * md1, (D // md1) * W, C)


def _unfold(y, d, h, w, md1=8):
    B, C = y.shape[0], y.shape[-1]
    return y.view(B, h, md1, d // md1, w, C).permute(0, 2, 3, 1, 4, 5).reshape(B, d, h, w, C)


def generator_forward_train(gen, x):
    """Residual_mid_UNet3D_vit.forward(x) (model.py:137-175, output_vit_mid=False) with autograd: x (B, 1, D, H, W) f32 -> (B, 1, D, H, W) f32."""
    if not x.is_cuda:
        raise RuntimeError("the MI355X generator runs on the GPU only (no CPU fallback)")
    x = x.contiguous().float()
    feats = []
    h = x
    for enc in gen.encoders:
        if enc.pooling is not None:
            h = _MaxPoolFn.apply(h)
        h = resnet_block(enc.basic_module, h)
        feats.insert(0, h)
    feats = feats[1:]
    d, hh, w = h.shape[1:4]
    mid_in = _fold(h).contiguous()
    mid_out = vit_forward_train(gen.mid, mid_in)
    h = _unfold(mid_out, d, hh, w).contiguous()
    for dec, ef in zip(gen.decoders, feats):
        up = dec.upsampling
        h = _UpJoinFn.apply(ef, h, up.upsample.conv_transposed.weight, up)
        h = resnet_block(dec.basic_module, h)
    pet = _Out1Fn.apply(h, gen.final_conv.weight, gen.final_conv.bias)
    if gen.final_activation is not None and not gen.training:
        pet = gen.final_activation(pet)
    return pet


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        p_, t_ = pred.detach().float().contiguous().reshape(-1), target.detach().float().contiguous().reshape(-1)
        assert p_.numel() == t_.numel() and p_.is_cuda, "l1_loss: HIP path only, equal sizes"
        n = p_.numel()
        loss = torch.empty(1, dtype=torch.float32, device=p_.device)
        dp = torch.empty_like(p_)
        ws = torch.empty(lib().gfe_l1_loss_blocks(n), dtype=torch.float32, device=p_.device)
        call("gfe_l1_loss", ptr(p_), ptr(t_), ptr(loss), ptr(dp), ptr(ws), n, stream())
        ctx.save_for_backward(dp)
        ctx.meta = (pred.shape, pred.dtype)
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        (dp,) = ctx.saved_tensors
        shape, dt = ctx.meta
        return (dp * dloss).view(shape).to(dt), None


def l1_loss(pred, target):
    """nn.L1Loss()(pred, target) (main_gan_vit.py:72): value and gradient from one launch, per-block partial sums added in order (bit-reproducible)."""
    return _L1Loss.apply(pred, target)


def train_step(gen, opt, x, target):
    """One generator step of main_gan_vit.py:68-82 without the third-party losses: L1(model(condition), real) -> backward -> optimizer."""
    opt.zero_grad()
    pred = generator_forward_train(gen, x)
    loss = l1_loss(pred, target)
    loss.backward()
    opt.step()
    return loss.detach()
