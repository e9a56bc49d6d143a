"""Building blocks of the residual 3D-UNet generator -- MI355X build.

Same class names, constructor arguments and state-dict keys as the reference's
pytorch3dunet/unet3d/buildingblocks.py (create_conv :10-86, SingleConv :89-115, ResNetBlock :180-229, Encoder :251-309,
Decoder :312-400, create_encoders/decoders :403-461, TransposeConvUpsampling :498-542), restricted to the path
`Residual_mid_UNet3D_vit` uses: ResNetBlock basic module, GroupNorm-before-conv orders ('gcr' / 'gc'), transposed-conv
upsampling with summation joining.  The torch layers below only HOLD parameters; forward() runs the HIP kernels of
libgfe_hip.so on channels-last bf16 activations (B, D, H, W, C).  Inference only (the generator is frozen:
classify_mamba.py:53,100).
"""
import torch
import torch.nn.functional as F
from torch import nn

from gfe_hip import nn_ops as K


class _PackCache:
    """Packed bf16 weights, rebuilt when the source parameters change (load_state_dict / .to() / a FlatAdam update, which rewrites the
    storage from a kernel without a version bump and therefore counts its steps in p._gfe_epoch)."""

    def __init__(self):
        self.sig, self.val = None, None

    def get(self, params, build):
        sig = tuple((p.data_ptr(), p._version, str(p.device), getattr(p, "_gfe_epoch", (0,))[0]) for p in params)
        if sig != self.sig:
            with torch.no_grad():
                self.val = build()
            self.sig = sig
        return self.val


def _f32(p):
    return p.detach().float().contiguous()


class SingleConv(nn.Sequential):
    """GroupNorm(in_channels) -> Conv3d(k3, p1, no bias) [-> ReLU]  (order 'gcr' / 'gc'; buildingblocks.py:38-67)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, order='gcr', num_groups=8, padding=1, dropout_prob=0.1, is3d=True):
        super().__init__()
        assert is3d and kernel_size == 3 and padding == 1, "MI355X build covers the 3-D k3 p1 path"
        assert order in ('gcr', 'gc'), f"layer order {order!r} is outside the hot path (GroupNorm before conv only)"
        if in_channels < num_groups:
            num_groups = 1                                           # buildingblocks.py:62-63
        assert in_channels % num_groups == 0
        self.add_module('groupnorm', nn.GroupNorm(num_groups=num_groups, num_channels=in_channels))
        self.add_module('conv', nn.Conv3d(in_channels, out_channels, 3, padding=1, bias=False))
        if 'r' in order:
            self.add_module('ReLU', nn.ReLU(inplace=True))
        self.relu = 'r' in order
        self._pack = _PackCache()

    def forward(self, x, residual=None, stats=False, out1=None):
        """x: (B, D, H, W, Cin) bf16; optional residual is added before the (forced) ReLU: ResNetBlock tail.
        stats: the output feeds another SingleConv -> its GroupNorm partials are produced by this conv's epilogue."""
        gn, conv = self.groupnorm, self.conv
        w32, g, b = self._pack.get([conv.weight, gn.weight, gn.bias],
                                   lambda: (K.pack_conv3(conv.weight, torch.float32), _f32(gn.weight), _f32(gn.bias)))
        scale, shift = K.groupnorm_scale_shift(x, g, b, gn.num_groups, gn.eps)
        # GroupNorm is folded into per-sample weights + a boundary-class bias table: the conv itself streams raw activations
        wb, tab = K.fold_groupnorm(w32, scale, shift, K.CONV3_TAPS, conv.in_channels, conv.out_channels)
        if out1 is not None:            # (w (C,), b): the generator's final 1x1x1 conv, fused into this conv's epilogue
            return K.conv3_out1(x, wb, tab, conv.out_channels, residual, out1[0], out1[1], relu=self.relu or residual is not None)
        return K.conv_igemm(x, wb, K.CONV3_TAPS, conv.out_channels, bias_tab=tab, res=residual, relu=self.relu or residual is not None,
                            stats=True if stats else None)


class ResNetBlock(nn.Module):
    """r = conv1(x); o = conv3(conv2(r)); relu(o + r)  (buildingblocks.py:218-229; ReLU because order has no 'l'/'e')."""

    def __init__(self, in_channels, out_channels, kernel_size=3, order='gcr', num_groups=8, is3d=True, **kwargs):
        super().__init__()
        self.conv1 = nn.Conv3d(in_channels, out_channels, 1) if in_channels != out_channels else nn.Identity()
        self.conv2 = SingleConv(out_channels, out_channels, kernel_size=kernel_size, order=order, num_groups=num_groups, is3d=is3d)
        n_order = order
        for c in 'rel':
            n_order = n_order.replace(c, '')
        self.conv3 = SingleConv(out_channels, out_channels, kernel_size=kernel_size, order=n_order, num_groups=num_groups, is3d=is3d)
        assert 'l' not in order and 'e' not in order
        self.non_linearity = nn.ReLU(inplace=True)
        self._pack = _PackCache()
        self._pack2 = _PackCache()

    def lift(self, x):
        """conv1: 1x1x1 conv + bias; accepts the raw (B, 1, D, H, W) volume when in_channels == 1."""
        c1 = self.conv1
        if isinstance(c1, nn.Identity):
            return x
        if c1.in_channels == 1:
            w, b = self._pack.get([c1.weight, c1.bias], lambda: (_f32(c1.weight).view(-1), _f32(c1.bias)))
            return K.conv_in1(x, w, b, stats=True)
        if K.conv1x1_ok(c1.in_channels, c1.out_channels) and x.dim() == 5 and x.shape[-1] == c1.in_channels:
            # the plain streaming product (conv1x1.hip): no halo tile staged for one tap
            w, b = self._pack.get([c1.weight, c1.bias], lambda: (c1.weight.detach().reshape(c1.out_channels, c1.in_channels).to(torch.bfloat16).contiguous(), _f32(c1.bias)))
            return K.conv1x1(x, w, b, stats=True)
        w, b = self._pack.get([c1.weight, c1.bias], lambda: (K.pack_conv1(c1.weight), _f32(c1.bias)))
        return K.conv_igemm(x, w, [(0, 0, 0)], c1.out_channels, bias=b, stats=True)

    def _first_block(self, x):
        """The whole block for a ONE-channel x without ever materialising r = conv1(x) = w1 x + b1 (906 MB at 96^3, B=8):
          * GroupNorm is affine per (sample, channel) and its statistics of r follow from the first two moments of x
            (gfe_lift_groupnorm_affine);
          * conv2(GroupNorm(r)) is then a one-channel 27-tap convolution of x with per-sample effective weights
            (gfe_conv3d_c1_k3: K = 27 taps on the matrix cores, 0.25 ms instead of the 1.5 ms 64 -> 64 conv);
          * conv3's residual r is recomputed from x in the conv epilogue (gfe_conv3d_k3_lift_residual)."""
        c1, sc2, sc3 = self.conv1, self.conv2, self.conv3
        gn2, gn3, cv3 = sc2.groupnorm, sc3.groupnorm, sc3.conv
        w_oct, w1, b1, g2, be2, w32_3, g3, be3 = self._pack2.get(
            [sc2.conv.weight, c1.weight, c1.bias, gn2.weight, gn2.bias, cv3.weight, gn3.weight, gn3.bias],
            lambda: (_f32(sc2.conv.weight).reshape(sc2.conv.out_channels, sc2.conv.in_channels, 27).contiguous(), _f32(c1.weight).view(-1),
                     _f32(c1.bias), _f32(gn2.weight), _f32(gn2.bias), K.pack_conv3(cv3.weight, torch.float32), _f32(gn3.weight), _f32(gn3.bias)))
        x = x.contiguous()
        scale, shift = K.lift_groupnorm_affine(x, w1, b1, g2, be2, gn2.num_groups, gn2.eps)
        weff, tab = K.conv_c1_k3_tables(w_oct, scale, shift, w1, b1)
        o = K.conv_c1_k3(x, weff, tab, relu=sc2.relu)                               # carries its GroupNorm partials
        s3, t3 = K.groupnorm_scale_shift(o, g3, be3, gn3.num_groups, gn3.eps)
        wb, tab3 = K.fold_groupnorm(w32_3, s3, t3, K.CONV3_TAPS, cv3.in_channels, cv3.out_channels)
        return K.conv3_lift_residual(o, wb, tab3, cv3.out_channels, x, w1, b1, relu=True)      # relu(conv3 + r): buildingblocks.py:226-229

    def _conv2_through_lift(self, x, r):
        """conv2(GroupNorm(conv1(x))) for a multi-channel x: conv1 is a 1x1x1 linear map Cin -> C (C = 2 Cin in the encoders) and
        GroupNorm is affine per (sample, channel), so the C -> C convolution equals a Cin -> C convolution of x itself with per-sample
        weights W_eff[o][tap][i] = sum_c W2[o][c][tap] s_c W1[c][i] (a 0.06 GFLOP matrix product per sample) and the boundary-class
        bias table of s_c b1_c + t_c: HALF the MFMA work and half the activation staging of conv2 (the encoders' 64 -> 128 and
        128 -> 256 levels)."""
        c1, sc = self.conv1, self.conv2
        gn, conv = sc.groupnorm, sc.conv
        cin, c, cout = c1.in_channels, c1.out_channels, conv.out_channels
        w32, w2m, w1, b1, g, b = self._pack2.get(
            [conv.weight, c1.weight, c1.bias, gn.weight, gn.bias],
            lambda: (lambda wp: (wp, wp.permute(1, 2, 0, 3).reshape(27 * wp.shape[2], -1)[:, :c].contiguous(),
                                 _f32(c1.weight).view(c, cin), _f32(c1.bias), _f32(gn.weight), _f32(gn.bias)))(K.pack_conv3(conv.weight, torch.float32)))
        scale, shift = K.groupnorm_scale_shift(r, g, b, gn.num_groups, gn.eps)                 # (B, C) from r's partials
        B, cp, nslab = scale.shape[0], w32.shape[2], (cin + 31) // 32
        # W_eff of all samples as ONE f32 MFMA GEMM (gfe_gemm_f32): (27*cp, C) . [s_b (.) W1]_b (C, B*Cin); its operands and the re-layout
        # of its result (per-sample packed bf16 weight sets (B, nslab, 27, cp, 32)) are one launch each (gfe_lift_fold_prep / _pack)
        rhs, shift2 = K.lift_fold_prep(scale, shift, w1, b1)
        weff = K.lift_fold_pack(K.gemm_f32(w2m, False, rhs, True), B, cin, cp)
        _, tab = K.fold_groupnorm(w32, scale, shift2, K.CONV3_TAPS, c, cout, weights=False)       # only the bias table is wanted
        return K.conv_igemm(x, weff, K.CONV3_TAPS, cout, bias_tab=tab, relu=sc.relu, stats=True)

    def forward(self, x, out1=None):
        """out1 = (w, b) of a following 1x1x1 conv C -> 1 (the generator's final_conv): returns that conv's output instead of the block's."""
        c1 = self.conv1
        if (not isinstance(c1, nn.Identity) and c1.in_channels == 1 and c1.out_channels == 64 and x.dim() == 5 and x.shape[1] == 1
                and x.dtype == torch.float32 and self.conv3.conv.out_channels == 64):
            return self._first_block(x)
        r = self.lift(x)                                  # r and o carry their GroupNorm partials (written by the producing kernel)
        lifted = not isinstance(c1, nn.Identity) and getattr(r, "gn_partials", None) is not None
        if lifted and c1.in_channels >= 32 and c1.in_channels % 8 == 0 and x.dim() == 5 and x.shape[-1] == c1.in_channels:
            o = self._conv2_through_lift(x, r)
        else:
            o = self.conv2(r, stats=True)
        if out1 is not None and isinstance(c1, nn.Identity) and self.conv3.conv.out_channels == 64:
            return self.conv3(o, residual=r, out1=out1)
        return self.conv3(o, residual=r)


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, conv_kernel_size=3, apply_pooling=True, pool_kernel_size=2, pool_type='max',
                 basic_module=ResNetBlock, conv_layer_order='gcr', num_groups=8, padding=1, upscale=2, dropout_prob=0.1, is3d=True):
        super().__init__()
        assert pool_type == 'max' and pool_kernel_size == 2 and is3d
        self.pooling = nn.MaxPool3d(kernel_size=2) if apply_pooling else None
        self.basic_module = basic_module(in_channels, out_channels, encoder=True, kernel_size=conv_kernel_size,
                                         order=conv_layer_order, num_groups=num_groups, padding=padding, upscale=upscale,
                                         dropout_prob=dropout_prob, is3d=is3d)

    def forward(self, x):
        if self.pooling is not None:
            tag = getattr(x, "pooled2", None)             # the producing conv already pooled its result in the epilogue (first block)
            x = tag[0] if (tag is not None and tag[1] == x._version) else K.maxpool2(x)
        return self.basic_module(x)


class TransposeConvUpsampling(nn.Module):
    """ConvTranspose3d(k3, s2, p1, no bias) then nearest resize to the encoder feature size (buildingblocks.py:498-542)."""

    class Upsample(nn.Module):
        def __init__(self, conv_transposed, is3d):
            super().__init__()
            self.conv_transposed = conv_transposed
            self.is3d = is3d

    def __init__(self, in_channels, out_channels, kernel_size=3, scale_factor=2, is3d=True):
        super().__init__()
        assert is3d and kernel_size == 3 and scale_factor == 2
        self.upsample = self.Upsample(nn.ConvTranspose3d(in_channels, out_channels, kernel_size=3, stride=2, padding=1, bias=False), is3d)
        self._pack = _PackCache()

    def forward(self, encoder_features, x):
        """Returns encoder_features + resize(convT(x)) -- the upsampling AND the summation join (buildingblocks.py:396-400)."""
        ct = self.upsample.conv_transposed
        classes, fused = self._pack.get([ct.weight], lambda: self._build(ct))
        out = torch.empty_like(encoder_features)
        if fused is not None:                              # one launch for the 8 parity classes (64-channel groups)
            return K.convT_fused(x, fused, ct.out_channels, encoder_features, out)
        B, D, H, W, _ = x.shape
        slots = K.conv_stat_slots(B, D, H, W, ct.out_channels)
        ws = K.new_gn_partials(B, 8 * slots, ct.out_channels, x.device)    # the 8 parity classes fill disjoint slot ranges
        for i, (par, (w, taps)) in enumerate(classes.items()):
            K.conv_igemm(x, w, taps, ct.out_channels, res=encoder_features, transposed=(par, out), stats=(ws, i * slots))
        assert len(classes) == 8 and K.cout_pad(ct.out_channels) < 64      # (wider ones take the fused launch above)
        out.gn_partials = ws
        return out

    @staticmethod
    def _build(ct):
        classes = K.pack_convT(ct.weight)
        fused = K.fuse_convT(classes) if K.cout_pad(ct.out_channels) >= 64 else None
        return classes, fused


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, conv_kernel_size=3, scale_factor=2, basic_module=ResNetBlock,
                 conv_layer_order='gcr', num_groups=8, padding=1, upsample='default', dropout_prob=0.1, is3d=True):
        super().__init__()
        assert basic_module is ResNetBlock and upsample in ('default', 'deconv'), "hot path = deconv upsampling + summation join"
        self.upsampling = TransposeConvUpsampling(in_channels, out_channels, kernel_size=conv_kernel_size, scale_factor=scale_factor, is3d=is3d)
        self.basic_module = basic_module(out_channels, out_channels, encoder=False, kernel_size=conv_kernel_size,
                                         order=conv_layer_order, num_groups=num_groups, padding=padding,
                                         dropout_prob=dropout_prob, is3d=is3d)

    def forward(self, encoder_features, x, out1=None):
        x = self.upsampling(encoder_features, x)          # upsample + nearest resize + sum join, one fused epilogue
        return self.basic_module(x, out1=out1) if out1 is not None else self.basic_module(x)

    def fuses_out1(self):
        """True when forward(..., out1=(w, b)) returns the final 1x1x1 conv's output (the 64-channel identity-lift block)."""
        bm = self.basic_module
        return isinstance(bm.conv1, nn.Identity) and bm.conv3.conv.out_channels == 64


def create_encoders(in_channels, f_maps, basic_module, conv_kernel_size, conv_padding, conv_upscale, dropout_prob,
                    layer_order, num_groups, pool_kernel_size, is3d):
    encoders = []
    for i, out_feature_num in enumerate(f_maps):
        encoders.append(Encoder(in_channels if i == 0 else f_maps[i - 1], out_feature_num, apply_pooling=i > 0,
                                basic_module=basic_module, conv_layer_order=layer_order, conv_kernel_size=conv_kernel_size,
                                num_groups=num_groups, pool_kernel_size=pool_kernel_size, padding=conv_padding,
                                upscale=conv_upscale, dropout_prob=dropout_prob, is3d=is3d))
    return nn.ModuleList(encoders)


def create_decoders(f_maps, basic_module, conv_kernel_size, conv_padding, layer_order, num_groups, upsample, dropout_prob, is3d):
    decoders = []
    rev = list(reversed(f_maps))
    for i in range(len(rev) - 1):
        decoders.append(Decoder(rev[i], rev[i + 1], basic_module=basic_module, conv_layer_order=layer_order,
                                conv_kernel_size=conv_kernel_size, num_groups=num_groups, padding=conv_padding,
                                upsample=upsample, dropout_prob=dropout_prob, is3d=is3d))
    return nn.ModuleList(decoders)
