"""Tensor-level wrappers of the generator / GEMM / ViT entry points of include/gfe_hip.h.

Pure plumbing: shape checks, output allocation (torch caching allocator), raw pointers + current stream.
Activations of the generator are channels-last bf16: (B, D, H, W, C).
"""
import ctypes
import os
import itertools

import numpy as np
import torch

from . import GFE_BF16, GFE_F32, call, dtype_code, lib, ptr, stream

BF16 = torch.bfloat16


def _i8(a):
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.int8))
    return arr, arr.ctypes.data


def _i64(a):
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.int64))
    return arr, arr.ctypes.data


def cout_pad(cout):
    return lib().gfe_conv3d_cout_pad(cout)


# ---- weight packing -----------------------------------------------------------------------------------------------
CONV3_TAPS = [(kd - 1, kh - 1, kw - 1) for kd in range(3) for kh in range(3) for kw in range(3)]


def _row_perm(cp):
    """Packed weight row -> output channel.  Inside a group of NT 16-row MFMA tiles, row ct*16 + 4*lq + r carries channel
    lq*4*NT + 4*ct + r, so that an MFMA lane (which owns rows 4*lq..4*lq+3 of every tile) ends up with 4*NT consecutive
    output channels of its voxel and the epilogue stores contiguous 8*NT-byte pieces."""
    nt = min(cp, 64) // 16
    rho = torch.arange(cp)
    g, r = rho // (nt * 16), rho % (nt * 16)
    return g * nt * 16 + ((r >> 2) & 3) * 4 * nt + (r >> 4) * 4 + (r & 3)


def _pack(w_tco, cin, dtype=BF16):
    """w_tco: (ntaps, Cout, Cin) f32 -> [nslab][ntaps][CoutPad][32] bf16 (or f32), zero padded, rows permuted (see _row_perm)."""
    ntaps, cout, _ = w_tco.shape
    nslab = (cin + 31) // 32
    cp = cout_pad(cout)
    buf = torch.zeros((ntaps, cp, nslab * 32), dtype=torch.float32, device=w_tco.device)
    buf[:, :cout, :cin] = w_tco
    buf = buf[:, _row_perm(cp).to(buf.device)]
    return buf.view(ntaps, cp, nslab, 32).permute(2, 0, 1, 3).contiguous().to(dtype)


def pack_conv3(weight, dtype=BF16):
    """nn.Conv3d weight (Cout, Cin, 3, 3, 3) -> packed, tap order = CONV3_TAPS (cross-correlation: offset = k - 1)."""
    cout, cin = weight.shape[:2]
    return _pack(weight.detach().float().permute(2, 3, 4, 0, 1).reshape(27, cout, cin), cin, dtype)


_TAPS_DEV = {}


def taps_on_device(taps, device):
    key = (tuple(taps), str(device))
    if key not in _TAPS_DEV:
        _TAPS_DEV[key] = torch.tensor(taps, dtype=torch.int8, device=device).contiguous()
    return _TAPS_DEV[key]


def fold_groupnorm(w_packed_f32, scale, shift, taps, cin, cout, weights=True):
    """GroupNorm affine (scale, shift: (B, Cin) f32) folded into the conv: per-sample bf16 weights + boundary-class bias table
    (weights=False: the table alone, for a caller that builds its per-sample weights itself)."""
    B = scale.shape[0]
    cp = cout_pad(cout)
    dev = scale.device
    wout = torch.empty((B,) + tuple(w_packed_f32.shape), dtype=BF16, device=dev) if weights else None
    T = torch.empty((B, (cin + 31) // 32, len(taps), cp), dtype=torch.float32, device=dev)      # per-slab partials of the bias fold
    tab = torch.empty((B, 64, cp), dtype=torch.float32, device=dev)
    call("gfe_conv3d_fold_groupnorm", ptr(w_packed_f32), ptr(scale), ptr(shift), ptr(wout), ptr(T), ptr(tab),
         ptr(taps_on_device(taps, dev)), B, cin, cout, len(taps), stream())
    return wout, tab


def lift_fold_prep(scale, shift, w1, b1):
    """Operands of ResNetBlock._conv2_through_lift in one launch: rhs (C, B*Cin) = scale_b[c] * w1[c][i], shift2 (B, C) = scale * b1 + shift."""
    B, C = scale.shape
    cin = w1.shape[1]
    rhs = torch.empty((C, B * cin), dtype=torch.float32, device=scale.device)
    shift2 = torch.empty((B, C), dtype=torch.float32, device=scale.device)
    call("gfe_lift_fold_prep", ptr(scale), ptr(shift), ptr(w1), ptr(b1), ptr(rhs), ptr(shift2), B, C, cin, stream())
    return rhs, shift2


def lift_fold_pack(weff, B, cin, cp, ntaps=27):
    """weff (ntaps*cp, B*Cin) f32 -> (B, nslab, ntaps, cp, 32) bf16: per-sample weight sets in conv_igemm's layout, one launch."""
    assert weff.dtype == torch.float32 and weff.is_contiguous() and tuple(weff.shape) == (ntaps * cp, B * cin)
    out = torch.empty((B, (cin + 31) // 32, ntaps, cp, 32), dtype=BF16, device=weff.device)
    call("gfe_lift_fold_pack", ptr(weff), ptr(out), B, cin, cp, ntaps, stream())
    return out


def pack_conv1(weight):
    """nn.Conv3d weight (Cout, Cin, 1, 1, 1) -> packed single tap."""
    cout, cin = weight.shape[:2]
    return _pack(weight.detach().float().reshape(1, cout, cin), cin)


# per axis: output parity 0 (o = 2i): kernel index 1 at input offset 0; parity 1 (o = 2i+1): k=2 at offset 0, k=0 at offset +1
_CT_AXIS = {0: [(0, 1)], 1: [(0, 2), (1, 0)]}


def conv1x1(x, w_bf16, bias, stats=True):
    """1x1x1 conv Cin -> Cout + bias of a channels-last (B, D, H, W, Cin) bf16 tensor (gfe_conv1x1: ResNetBlock.conv1,
    buildingblocks.py:204-208).  w_bf16: the Conv3d weight as (Cout, Cin) bf16.  stats: tag the result with the GroupNorm partials
    of what was stored (`out.gn_partials`), as conv_igemm(stats=True) does."""
    B, D, H, W, cin = x.shape
    cout = w_bf16.shape[0]
    assert x.dtype == BF16 and x.is_contiguous() and w_bf16.dtype == BF16 and w_bf16.is_contiguous() and w_bf16.shape[1] == cin
    V = D * H * W
    out = torch.empty((B, D, H, W, cout), dtype=BF16, device=x.device)
    ws = None
    if stats:
        ws = torch.empty((B, lib().gfe_conv1x1_stat_slots(V), 2, cout), dtype=torch.float32, device=x.device)     # every element is written
    call("gfe_conv1x1", ptr(x), ptr(w_bf16), ptr(bias), ptr(out), B, V, cin, cout, ptr(ws), 0 if ws is None else ws.shape[1], 0, stream())
    if stats:
        out.gn_partials = ws
    return out


def conv1x1_ok(cin, cout):
    """Shapes gfe_conv1x1 takes (everything else stays on the one-tap conv_igemm path)."""
    if cin not in (64, 128) or cout < 64 or cout % 64:
        return False
    g = cout // (128 if (cin == 64 and cout % 128 == 0) else 64)
    return g in (1, 2, 4)


def pack_convT(weight):
    """nn.ConvTranspose3d weight (Cin, Cout, 3, 3, 3), stride 2, padding 1 -> {parity (pd,ph,pw): (packed, taps)}."""
    cin, cout = weight.shape[:2]
    w = weight.detach().float()
    out = {}
    for par in itertools.product((0, 1), repeat=3):
        taps, mats = [], []
        for (od, kd), (oh, kh), (ow, kw) in itertools.product(_CT_AXIS[par[0]], _CT_AXIS[par[1]], _CT_AXIS[par[2]]):
            taps.append((od, oh, ow))
            mats.append(w[:, :, kd, kh, kw].t())            # (Cout, Cin)
        out[par] = (_pack(torch.stack(mats, 0), cin), taps)
    return out


def fuse_convT(classes):
    """The 8 per-class packs of pack_convT as one buffer + the host tables gfe_convt3d_k3s2_fused takes: heavy classes first so
    that the tail of a persistent block's work list is made of light items."""
    order = sorted(classes.keys(), key=lambda par: -len(classes[par][1]))
    parts, woff, ntaps, parity, taps, off = [], [], [], [], [], 0
    for par in order:
        w, tl = classes[par]
        parts.append(w.reshape(-1))
        woff.append(off)
        off += w.numel()
        ntaps.append(len(tl))
        parity.extend(par)
        taps.extend(tl)
    return dict(w=torch.cat(parts).contiguous(), woff=_i64(woff), ntaps=(np.ascontiguousarray(np.asarray(ntaps, dtype=np.int32)),),
                parity=_i8(parity), taps=_i8(taps), elems=off)


def convT_fused(x, fused, cout, res, out, stats=True):
    """ConvTranspose3d(k3, s2, p1) + nearest resize to out's size + `res + .` in one launch (gfe_convt3d_k3s2_fused); out is
    tagged with the GroupNorm partials of the result."""
    B, D, H, W, cin = x.shape
    OD, OH, OW = out.shape[1:4]
    oshift = OD - (2 * D - 1)
    assert oshift in (0, 1) and OH == 2 * H - 1 + oshift and OW == 2 * W - 1 + oshift, "unsupported upsampling size"
    stats = stats and cout % 64 == 0          # 64-channel tiles write 8-channel sums (gfe_hip.h)
    ws = new_gn_partials(B, lib().gfe_convt3d_stat_slots(B, D, H, W, cout), cout, x.device) if stats else None
    nt = fused["ntaps"][0]
    call("gfe_convt3d_k3s2_fused", ptr(x), ptr(fused["w"]), fused["woff"][1], nt.ctypes.data, fused["parity"][1], fused["taps"][1],
         fused["elems"], ptr(res), ptr(out), B, D, H, W, cin, cout, OD, OH, OW, oshift, ptr(ws), 0 if ws is None else ws.shape[1], stream())
    if stats:
        out.gn_partials = ws
    return out


# ---- generator ops --------------------------------------------------------------------------------------------------
def conv_tiles(D, H, W):
    return lib().gfe_conv3d_tiles(D, H, W)


def conv_stat_slots(B, D, H, W, cout):
    """GroupNorm-partial slots per sample that one conv_igemm(stats=...) call writes (gfe_conv3d_stat_slots)."""
    return lib().gfe_conv3d_stat_slots(B, D, H, W, cout)


def new_gn_partials(B, nblk, C, device):
    """Zeroed workspace a producer fills with the GroupNorm partials of its output (gfe_hip.h: stats_ws): producers that do not write every
    slot (the transposed conv's two tile grids, callers that hand several launches one workspace) rely on the rest reading as zero."""
    return torch.zeros((B, nblk, 2, C), dtype=torch.float32, device=device)


def conv_igemm(x, w_packed, taps, cout, bias=None, bias_tab=None, res=None, relu=False, out=None, transposed=None, stats=None):
    """x: (B, D, H, W, Cin) bf16.  w_packed: one weight set, or (B, ...) per-sample sets from fold_groupnorm (with bias_tab).
    transposed: None, or (parity tuple, out tensor (B, 2D, 2H, 2W, Cout)) for one ConvT class.
    stats: None | True | (ws, slot0): write the GroupNorm partials of the output; True allocates them and tags the result
    (`out.gn_partials`) so the next SingleConv skips its statistics pass."""
    B, D, H, W, cin = x.shape
    assert x.dtype == BF16 and x.is_contiguous()
    tarr, tptr = _i8(taps)
    if transposed is None:
        if out is None:
            out = torch.empty((B, D, H, W, cout), dtype=BF16, device=x.device)
        OD, OH, OW = D, H, W
        ostride, par, oshift = 1, (0, 0, 0), 0
    else:
        par, out = transposed
        OD, OH, OW = out.shape[1:4]
        ostride = 2
        oshift = OD - (2 * D - 1)
        assert oshift in (0, 1) and OH == 2 * H - 1 + oshift and OW == 2 * W - 1 + oshift, "unsupported upsampling size"
    wstride = w_packed.stride(0) if w_packed.dim() == 5 else 0
    ws, slot0 = None, 0
    if stats is True and cout >= 64 and cout % 64:
        stats = None                 # 64-channel tiles write 8-channel sums: other widths take the separate statistics pass
    if stats is True:
        # a stride-1 launch writes EVERY slot it is given -- one per tile of the sample, all channels, a plain store when the block leaves the tile
        # (conv3d.hip, "flush") -- so the workspace needs no zero fill (a 5-10 us launch in front of six convs of a generator pass);
        # tools/poison_probe.py screens exactly this kind of claim
        nslot = conv_stat_slots(B, D, H, W, cout)
        ws = torch.empty((B, nslot, 2, cout), dtype=torch.float32, device=x.device) if transposed is None else new_gn_partials(B, nslot, cout, x.device)
    elif stats is not None:
        ws, slot0 = stats
    call("gfe_conv3d_igemm", ptr(x), ptr(w_packed), wstride, ptr(bias), ptr(bias_tab), ptr(res), ptr(out),
         B, D, H, W, cin, cout, OD, OH, OW, len(taps), tptr, ostride, par[0], par[1], par[2], oshift, int(relu),
         ptr(ws), 0 if ws is None else ws.shape[1], slot0, stream())
    if stats is True:
        out.gn_partials = ws
    return out


def groupnorm_scale_shift(x, gamma, beta, groups, eps=1e-5):
    """x: (B, ..., C) bf16 channels-last -> (scale, shift) each (B, C) f32."""
    B, C = x.shape[0], x.shape[-1]
    S = x.numel() // (B * C)
    part = getattr(x, "gn_partials", None)
    if part is not None:                         # the producer of x already reduced it tile by tile
        nblk = part.shape[1]
        ss = torch.empty((2, B, C), dtype=torch.float32, device=x.device)
        ws2 = torch.empty((B, 32, 2, C), dtype=torch.float32, device=x.device) if nblk > 512 else None
        call("gfe_groupnorm_from_partials", ptr(part), nblk, ptr(gamma), ptr(beta), ptr(ss[0]), ptr(ss[1]), ptr(ws2), B, S, C, groups, eps, stream())
        return ss[0], ss[1]
    vpb, nblk = ctypes.c_int(), ctypes.c_int()
    lib().gfe_groupnorm_plan(S, ctypes.byref(vpb), ctypes.byref(nblk))
    ws = torch.empty((B, nblk.value, 2, C), dtype=torch.float32, device=x.device)
    ss = torch.empty((2, B, C), dtype=torch.float32, device=x.device)
    call("gfe_groupnorm_scale_shift", ptr(x), ptr(gamma), ptr(beta), ptr(ss[0]), ptr(ss[1]), ptr(ws), B, S, C, groups, eps, stream())
    return ss[0], ss[1]


_C1_MASK = {}


def conv_c1_k3_tables(w32_oct, scale, shift, w1, b1):
    """Effective weights / bias table of conv(GroupNorm(w1 x + b1)) as a one-channel 27-tap conv of x (gfe_conv3d_c1_k3):
    w32_oct: (Cout, Cin, 27) f32 conv weight, taps in CONV3_TAPS order; scale, shift: (B, Cin) GroupNorm affine of r = w1 x + b1;
    w1, b1: (Cin,) the 1x1x1 lift.  Returns weff (B, 27, Cout), tab (B, 64, Cout)."""
    dev = w32_oct.device
    m = _C1_MASK.get(str(dev))
    if m is None:                                   # mask[cls][tap] = 1 when the tap stays inside the volume for boundary class cls
        rows = []
        for cls in range(64):
            rows.append([0.0 if ((dd < 0 and cls & 1) or (dd > 0 and cls & 2) or (dh < 0 and cls & 4) or (dh > 0 and cls & 8) or
                                 (dw < 0 and cls & 16) or (dw > 0 and cls & 32)) else 1.0 for dd, dh, dw in CONV3_TAPS])
        m = _C1_MASK[str(dev)] = torch.tensor(rows, dtype=torch.float32, device=dev)
    sw = scale * w1                                                   # (B, Cin)
    sh = scale * b1 + shift
    # weight-only factors, cached: Wt[(t, o)][c] = W[o][c][t] and Wm[(cls, o)][c] = sum_t mask[cls][t] W[o][c][t]; the per-sample tables
    # are then two small f32 MFMA GEMMs (gfe_gemm_f32), no library GEMM on the path
    key = (w32_oct, w32_oct._version)               # identity, not address: a rebuilt pack may land on the freed one's address
    ent = _C1_WT.get(str(dev))
    if ent is None or ent[0][0] is not key[0] or ent[0][1] != key[1]:
        cout, cin = w32_oct.shape[0], w32_oct.shape[1]
        wt = w32_oct.permute(2, 0, 1).reshape(27 * cout, cin).contiguous()
        wm = gemm_f32(m, False, w32_oct.permute(2, 0, 1).reshape(27, cout * cin).contiguous(), True).reshape(64 * cout, cin)
        ent = _C1_WT[str(dev)] = (key, wt, wm)
    _, wt, wm = ent
    B, cout = scale.shape[0], w32_oct.shape[0]
    return gemm_f32(sw, False, wt, False).view(B, 27, cout), gemm_f32(sh, False, wm, False).view(B, 64, cout)


_C1_WT = {}


def lift_groupnorm_affine(x, w1, b1, gamma, beta, groups, eps=1e-5):
    """GroupNorm scale / shift (B, C) of r = w1 x + b1 for a one-channel volume x (B, 1, D, H, W) f32, from the moments of x."""
    B = x.shape[0]
    S = x.numel() // B
    C = w1.numel()
    ss = torch.empty((2, B, C), dtype=torch.float32, device=x.device)
    ws = torch.empty(B * 128, dtype=torch.float64, device=x.device)
    call("gfe_lift_groupnorm_affine", ptr(x), ptr(w1), ptr(b1), ptr(gamma), ptr(beta), ptr(ss[0]), ptr(ss[1]), ptr(ws), B, S, C, groups, eps, stream())
    return ss[0], ss[1]


def conv3_lift_residual(x, w_packed, tab, cout, vol, lift_w, lift_b, relu=True, pool=True):
    """27-tap conv of x (B, D, H, W, Cin) bf16 with per-sample folded weights + bias table, plus the residual w[c] * vol + b[c]
    recomputed in the epilogue from the one-channel volume (gfe_conv3d_k3_lift_residual).  pool: the epilogue also writes
    MaxPool3d(2) of the result (attached as `out.pooled2 = (tensor, out._version)`; the next Encoder takes it instead of re-reading the
    tensor, provided `out` has not been written since -- an in-place edit bumps the version and the pooling is redone; a re-wrapped tensor
    has no attribute and is pooled the ordinary way)."""
    B, D, H, W, cin = x.shape
    assert x.dtype == BF16 and x.is_contiguous() and vol.dtype == torch.float32 and vol.is_contiguous() and w_packed.dim() == 5
    out = torch.empty((B, D, H, W, cout), dtype=BF16, device=x.device)
    pool = pool and relu and D % 2 == 0 and H % 2 == 0 and W % 2 == 0 and os.environ.get("GFE_NO_FUSED_POOL") != "1"
    pooled = torch.empty((B, D // 2, H // 2, W // 2, cout), dtype=BF16, device=x.device) if pool else None
    _, tptr = _i8(CONV3_TAPS)
    call("gfe_conv3d_k3_lift_residual", ptr(x), ptr(w_packed), w_packed.stride(0), ptr(tab), ptr(out), B, D, H, W, cin, cout, tptr, int(relu),
         ptr(vol), ptr(lift_w), ptr(lift_b), ptr(pooled), stream())
    if pooled is not None:
        out.pooled2 = (pooled, out._version)      # valid only for THIS content of `out`: the consumer checks the version counter
    return out


def conv3_out1(x, w_packed, tab, cout, res, out_w, out_b, relu=True):
    """27-tap conv of x (per-sample folded weights + bias table) + bf16 residual + ReLU, followed by the final 1x1x1 conv cout -> 1:
    returns (B, 1, D, H, W) f32; the 64-channel tensor in between is never stored (gfe_conv3d_k3_out1)."""
    B, D, H, W, cin = x.shape
    assert x.dtype == BF16 and x.is_contiguous() and w_packed.dim() == 5 and (res is None or (res.dtype == BF16 and res.is_contiguous()))
    y = torch.empty((B, 1, D, H, W), dtype=torch.float32, device=x.device)
    _, tptr = _i8(CONV3_TAPS)
    call("gfe_conv3d_k3_out1", ptr(x), ptr(w_packed), w_packed.stride(0), ptr(tab), ptr(res), B, D, H, W, cin, cout, tptr, int(relu),
         ptr(out_w), float(out_b), ptr(y), stream())
    return y


def conv_c1_k3(x, weff, tab, relu=True):
    """x: (B, 1, D, H, W) f32|bf16 -> (B, D, H, W, 64) bf16 with its GroupNorm partials attached (`y.gn_partials`)."""
    B, _, D, H, W = x.shape
    C = weff.shape[-1]
    y = torch.empty((B, D, H, W, C), dtype=BF16, device=x.device)
    nblk = lib().gfe_conv3d_c1_k3_nblk(B, D, H, W)
    ws = torch.empty((B, nblk, 2, C), dtype=torch.float32, device=x.device)
    call("gfe_conv3d_c1_k3", ptr(x), ptr(weff), ptr(tab), ptr(y), ptr(ws), nblk, B, D, H, W, C, dtype_code(x.dtype), int(relu), stream())
    y.gn_partials = ws
    return y


def maxpool2(x):
    B, D, H, W, C = x.shape
    y = torch.empty((B, D // 2, H // 2, W // 2, C), dtype=BF16, device=x.device)
    call("gfe_maxpool3d_2", ptr(x), ptr(y), B, D, H, W, C, stream())
    return y


def conv_in1(x, w, bias, stats=False):
    """x: (B, 1, D, H, W) f32|bf16 contiguous -> (B, D, H, W, C) bf16; stats: also the GroupNorm partials (`y.gn_partials`)."""
    B, _, D, H, W = x.shape
    C = w.numel()
    y = torch.empty((B, D, H, W, C), dtype=BF16, device=x.device)
    if stats and 256 % (C // 8) == 0 and C <= 512:
        S = D * H * W
        ws = new_gn_partials(B, lib().gfe_conv_in1_nblk(S), C, x.device)
        call("gfe_conv_in1_stats", ptr(x), ptr(w), ptr(bias), ptr(y), ptr(ws), B, S, C, dtype_code(x.dtype), stream())
        y.gn_partials = ws
        return y
    call("gfe_conv_in1", ptr(x), ptr(w), ptr(bias), ptr(y), B * D * H * W, C, dtype_code(x.dtype), stream())
    return y


def conv_out1(x, w, bias_value):
    """x: (B, D, H, W, C) bf16 -> (B, 1, D, H, W) f32."""
    B, D, H, W, C = x.shape
    y = torch.empty((B, 1, D, H, W), dtype=torch.float32, device=x.device)
    call("gfe_conv_out1", ptr(x), ptr(w), float(bias_value), ptr(y), B * D * H * W, C, stream())
    return y


def fold_mid(x, md1=8, inverse=False, shape=None):
    """forward: (B, D, H, W, C) -> (B, H*md1, (D/md1)*W, C); inverse: the opposite, `shape` = (D, H, W)."""
    if not inverse:
        B, D, H, W, C = x.shape
        y = torch.empty((B, H * md1, (D // md1) * W, C), dtype=BF16, device=x.device)
    else:
        D, H, W = shape
        B, C = x.shape[0], x.shape[-1]
        y = torch.empty((B, D, H, W, C), dtype=BF16, device=x.device)
    call("gfe_fold_mid", ptr(x), ptr(y), B, D, H, W, C, md1, int(inverse), stream())
    return y


# ---- GEMM / ViT ops ---------------------------------------------------------------------------------------------------
def _auto_split_k(M, N, K):
    """few output tiles and a long K: the launch would occupy a fraction of the 256 CUs -> cut K across blocks (partials + a fixed-order sum).
    Up to 512 rows the cut is a function of (N, K) ALONE: the K partition decides how a row's sum is rounded, and a sample's rows must come
    out the same whatever batch they ride in (the generator's ViT runs 25 rows per volume: batch 1 .. 20 share one partition)."""
    if M <= 512:
        blocks = -(-N // 128)
        return max(1, min(K // 128, 128 // blocks)) if blocks < 64 else 1
    bm = 64 if (M % 128 != 0 and M % 128 <= 64 and M < 1024) else 128
    blocks = -(-M // bm) * -(-N // 128)
    return max(1, min(K // 128, 256 // blocks)) if blocks < 128 else 1


def gemm_nt(a, b, bias=None, res=None, act=0, out_dtype=BF16, split_k=1, out=None):
    """a: (M, K) bf16 (row stride allowed), b: (N, K) bf16 -> (M, N).  y = act(a b^T + bias) + res."""
    M, K = a.shape
    N = b.shape[0]
    assert a.dtype == BF16 and b.dtype == BF16 and a.stride(1) == 1 and b.stride(1) == 1 and b.shape[1] == K
    plain = out_dtype == torch.float32 and res is None and act == 0
    if split_k == 1 and out is None and K >= (1024 if plain else 256) and (plain or os.environ.get("GFE_GEMM_EPI_SPLIT", "1") == "1"):
        # skinny GEMMs (the generator ViT's M = 256 rows: 8-32 blocks walking K serially, latency-bound) are cut along K; with the ranges'
        # tiles in a workspace the epilogue (bias, GELU, residual, bf16 out) moves into the fixed-order reduction, so GEMMs with an
        # epilogue can be cut too: ViT GEMM time 0.67 -> 0.40 + 0.08 ms per step, the step 724-728 -> 730-733 volumes/s
        split_k = _auto_split_k(M, N, K)
        if not plain and (N % 4 or split_k * M * N * 4 > (64 << 20)):
            split_k = 1
    if out is None:
        out = (torch.zeros if split_k > 1 and plain else torch.empty)((M, N), dtype=out_dtype, device=a.device)
    call("gfe_gemm_bf16_nt", ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0), M, N, K, ptr(bias),
         ptr(res), 0 if res is None else res.stride(0), int(res is not None and res.dtype == torch.float32),
         act, int(out.dtype == torch.float32), split_k, ptr(_splitk_ws(split_k, M, N, out)), stream())
    return out


def _splitk_ws(split_k, M, N, out):
    """Workspace of the deterministic split-K (range partials, summed in a fixed order): bit-reproducible results where f32 atomics
    differ from run to run.  None for a single range, or where the row stride of C rules out the 16-byte reduction."""
    if split_k <= 1 or out.stride(0) % 4:
        return None
    return torch.empty(split_k * M * N, dtype=torch.float32, device=out.device)


def gemm_ex(a, a_t, b, b_t, bias=None, out_dtype=torch.float32, split_k=1, accum_into=None):
    """C (M, N) = op(a) op(b)^T with bf16|f32 operands read in place (gfe_gemm_ex): `a_t` / `b_t` say that the operand is stored
    reduction-major, i.e. a is (K, M) / b is (K, N) in memory.  accum_into: f32 (M, N) tensor the product is ADDED to (a
    gradient buffer): through the residual epilogue, or directly by the split-K atomics."""
    M, K = (a.shape[1], a.shape[0]) if a_t else a.shape
    N = b.shape[1] if b_t else b.shape[0]
    assert (b.shape[0] if b_t else b.shape[1]) == K and a.stride(1) == 1 and b.stride(1) == 1
    mode = lambda t, tr: int(t.dtype == torch.float32) | (2 if tr else 0)
    assert a.dtype in (BF16, torch.float32) and b.dtype in (BF16, torch.float32)
    if split_k == 1 and out_dtype == torch.float32 and K >= 1024:
        split_k = _auto_split_k(M, N, K)
    res = None
    if accum_into is not None:
        assert accum_into.dtype == torch.float32 and accum_into.shape == (M, N) and accum_into.stride(1) == 1 and bias is None
        out = accum_into
        res = None if split_k > 1 else accum_into
    else:
        out = (torch.zeros if split_k > 1 else torch.empty)((M, N), dtype=out_dtype, device=a.device)
    call("gfe_gemm_ex", ptr(a), a.stride(0), mode(a, a_t), ptr(b), b.stride(0), mode(b, b_t), ptr(out), out.stride(0), M, N, K,
         ptr(bias), ptr(res), 0 if res is None else res.stride(0), int(res is not None), 0, int(out.dtype == torch.float32), split_k,
         ptr(_splitk_ws(split_k, M, N, out)), stream())
    return out


def gemm_f32(a, a_t, b, b_t, bias=None, accum_into=None):
    """Exact-f32 C (M, N) = op(a) op(b)^T (+ bias) on the f32 matrix cores (gfe_gemm_f32): a_t / b_t say that the operand is stored
    reduction-major, i.e. a is (K, M) / b is (K, N) in memory.  accum_into: f32 (M, N) gradient buffer the product is ADDED to.
    Any sizes and alignments (unit inner stride)."""
    M, K = (a.shape[1], a.shape[0]) if a_t else a.shape
    N = b.shape[1] if b_t else b.shape[0]
    unit = lambda t: t.shape[1] == 1 or t.stride(1) == 1          # (a size-1 dimension may carry any stride)
    ld = lambda t: t.stride(0) if t.shape[0] > 1 else t.shape[1]
    assert (b.shape[0] if b_t else b.shape[1]) == K and unit(a) and unit(b)
    assert a.dtype == torch.float32 and b.dtype == torch.float32
    # launch-bound sizes: the kernel uses 32 x 32 x 128 tiles below 512 blocks of 64 x 64; a block's time is its number of 128-deep
    # k-steps (one exposed memory round trip + 32 f32 MFMAs each), so K is split -- range partials + a fixed-order sum, bit-reproducible --
    # until ~1024 blocks are in flight, keeping >= 2 k-steps per block and the partial traffic (output bytes x splits) below ~8 MB
    # The head's own shapes (aligned operands, K-major a, K % 16 == 0) never get here: gfe_gemm_f32_inblock says the kernel cuts K between the
    # waves of ONE block per output tile and sums them in LDS (no reduction launch, no workspace).
    if lib().gfe_gemm_f32_inblock(ptr(a), ld(a), int(a_t), ptr(b), ld(b), int(b_t), M, N, K):
        split_k = 1
    else:
        blocks32 = -(-M // 32) * -(-N // 32)
        split_k = max(1, min(8, K // 256, 1024 // blocks32, (8 << 20) // max(1, 4 * M * N)))
        if os.environ.get("GFE_F32_SPLITK_MAX"):            # experiments: cap the cut (1 = no split-K, no reduction launch)
            split_k = min(split_k, int(os.environ["GFE_F32_SPLITK_MAX"]))
    if accum_into is not None:
        assert accum_into.dtype == torch.float32 and accum_into.shape == (M, N) and unit(accum_into) and bias is None
        out = accum_into
    else:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    ws = torch.empty(split_k * M * N, dtype=torch.float32, device=a.device) if split_k > 1 else None
    call("gfe_gemm_f32", ptr(a), ld(a), int(a_t), ptr(b), ld(b), int(b_t), ptr(out), ld(out), M, N, K,
         ptr(bias), int(accum_into is not None), split_k, ptr(ws), stream())
    return out


def _ex_ok(t):
    """operand usable in place by gemm_ex: 2-D, unit inner stride, 16-byte aligned rows"""
    al = 4 if t.dtype == torch.float32 else 8
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % al == 0 and t.data_ptr() % 16 == 0 and t.dtype in (BF16, torch.float32)


def transpose_bf16(x):
    """(..., R, C) bf16 contiguous -> (..., C, R)."""
    R, C = x.shape[-2:]
    batch = x.numel() // (R * C)
    y = torch.empty(x.shape[:-2] + (C, R), dtype=BF16, device=x.device)
    call("gfe_transpose_bf16", ptr(x), ptr(y), batch, R, C, C, R, stream())
    return y


def cast(x, dtype):
    x = x.contiguous()
    if x.dtype == dtype:
        return x
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    call("gfe_cast", ptr(x), ptr(y), x.numel(), int(dtype == BF16), stream())
    return y


def contig_map(rows, length):
    return [0, length, 0, 0, rows, 1, 1, length]


def patch_map(Himg, Wimg, C, p):
    """Row map of 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (vit.py:96) on a channels-last (B, Himg, Wimg, C) image."""
    return [Himg * Wimg * C, p * Wimg * C, p * C, Wimg * C, (Himg // p) * (Wimg // p), Wimg // p, p, p * C]


def layernorm(x, gamma, beta, rows, length, out_dtype, in_map=None, out_map=None, out=None, out_shape=None, eps=1e-5):
    im, ip = _i64(in_map or contig_map(rows, length))
    om, op = _i64(out_map or contig_map(rows, length))
    if out is None:
        out = torch.empty(out_shape or (rows, length), dtype=out_dtype, device=x.device)
    call("gfe_layernorm", ptr(x), ptr(out), ptr(gamma), ptr(beta), ip, op, rows, eps, dtype_code(x.dtype), dtype_code(out.dtype), stream())
    return out


def attention_fwd(q, k, v, B, H, n, dh, scale, with_lse=False, dropout_p=0.0, seed=0):
    """Flash-style attention for long sequences (gfe_attention_fwd): q/k/v (B*n, >= H*dh) bf16 views with a common layout
    (row strides allowed), dh == 64 -> (B*n, H*dh) bf16.  with_lse: also the (B, H, npad) f32 row statistic -(max + log2 sum) that
    attention_bwd restarts from (gfe_attention_fwd_lse); with it, dropout_p / seed: dropout on the probabilities (vit_3d.py:56; the mask is a
    hash of (seed, head, row, key) that attention_bwd regenerates from the same two numbers)."""
    assert with_lse or dropout_p == 0.0
    assert q.dtype == BF16 and k.dtype == BF16 and v.dtype == BF16 and q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
    o = torch.empty((B * n, H * dh), dtype=BF16, device=q.device)
    strides = (n * q.stride(0), q.stride(0), n * k.stride(0), k.stride(0), n * v.stride(0), v.stride(0), n * H * dh, H * dh)
    if not with_lse:
        call("gfe_attention_fwd", ptr(q), ptr(k), ptr(v), ptr(o), B, H, n, dh, *strides, float(scale), stream())
        return o
    nlse = torch.empty((B, H, -(-n // 64) * 64), dtype=torch.float32, device=q.device)
    call("gfe_attention_fwd_lse", ptr(q), ptr(k), ptr(v), ptr(o), ptr(nlse), B, H, n, dh, *strides, float(scale), float(dropout_p), int(seed), stream())
    return o, nlse


def attention_bwd(q, k, v, o, dout, nlse, B, H, n, dh, scale, dqkv=None, dropout_p=0.0, seed=0):
    """gfe_attention_bwd: gradients of attention_fwd(with_lse=True).  q/k/v: bf16 views with ONE common layout (slices of a (B*n, 3*H*dh)
    projection output), o/dout: (B*n, H*dh) bf16 -> (dq, dk, dv) bf16 views of one (B*n, 3*H*dh) buffer (`dqkv`, allocated if None): the
    layout the to_qkv weight-gradient GEMM consumes.  Deterministic (no atomics)."""
    inner = H * dh
    assert q.stride() == k.stride() == v.stride() and q.stride(1) == 1 and o.is_contiguous() and dout.is_contiguous() and dout.dtype == BF16
    if dqkv is None:
        dqkv = torch.empty((B * n, 3 * inner), dtype=BF16, device=q.device)
    dq, dk, dv = dqkv[:, :inner], dqkv[:, inner:2 * inner], dqkv[:, 2 * inner:]
    npad = nlse.shape[-1]
    qs = torch.empty((B * H * npad, dh), dtype=BF16, device=q.device)
    ndelta = torch.empty((B * H * npad,), dtype=torch.float32, device=q.device)
    call("gfe_attention_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(dout), ptr(nlse), ptr(dq), ptr(dk), ptr(dv), ptr(qs), ptr(ndelta),
         B, H, n, dh, n * q.stride(0), q.stride(0), n * inner, inner, n * dqkv.stride(0), dqkv.stride(0), float(scale), float(dropout_p), int(seed), stream())
    return dq, dk, dv


def attention_small(q, k, v, B, H, nq, nk, dh, scale):
    """q: (B*nq, >=H*dh) view, k/v: (B*nk, ...) views (row strides allowed) -> (B*nq, H*dh) bf16."""
    o = torch.empty((B * nq, H * dh), dtype=BF16, device=q.device)
    call("gfe_attention_small", ptr(q), ptr(k), ptr(v), ptr(o), B, H, nq, nk, dh,
         nq * q.stride(0), q.stride(0), nk * k.stride(0), k.stride(0), nk * v.stride(0), v.stride(0), nq * H * dh, H * dh,
         float(scale), stream())
    return o


def token_mix(x, w, bias, B, nin, nout, dim):
    y = torch.empty((B, nout, dim), dtype=BF16, device=x.device)
    call("gfe_token_mix", ptr(x), ptr(w), ptr(bias), ptr(y), B, nin, nout, dim, dtype_code(x.dtype), stream())
    return y


def vit_embed(tok, cls, pos, B, n, dim):
    x = torch.empty((B, n + 1, dim), dtype=torch.float32, device=tok.device)
    call("gfe_vit_embed", ptr(tok), ptr(cls), ptr(pos), ptr(x), B, n, dim, stream())
    return x
