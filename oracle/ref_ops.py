"""ORACLE -- test infrastructure, not product code.

A CPU restatement (plain torch ops on CPU tensors, fp32 or fp64) of the reference's algorithm for the
classify_mamba hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this; the product path (gfe-mamba_amd/) never does.

Pinning: the reference has no tests or golden vectors of its own.  This oracle is
pinned against outputs of the reference itself, generated in the build container by tools/make_golden.py
(which imports /root/reference) and committed as fixtures under tests/golden/; tests/test_oracle_golden.py
checks every function below against them.

All functions are pure: `sd` is a {state-dict key: tensor} mapping using the reference's key names, `pre`
a key prefix.  Citations are file:line in the reference repository.
"""
import math

import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------------
# Group A -- scan, Mamba
# ------------------------------------------------------------------------------------------------


def npo2(n):
    """cross_atten/pscan.py:13-18."""
    return 2 ** math.ceil(math.log2(n))


def pscan(A, X):
    """H[t] = A[t]*H[t-1] + X[t], H[-1] = 0, over dim 1 of (B, L, D, N).

    Sequential statement of what PScan.forward computes with its Blelloch sweeps (cross_atten/pscan.py:36-92,
    151-186; the recurrence is stated at pscan.py:41-44).  Padding to npo2(L) (pscan.py:20-33) does not change
    the first L outputs.
    """
    H = torch.empty_like(X)
    h = torch.zeros_like(X[:, 0])
    for t in range(X.shape[1]):
        h = A[:, t] * h + X[:, t]
        H[:, t] = h
    return H


def pscan_grads(A, H, gH):
    """PScan.backward (cross_atten/pscan.py:188-224): gX = reverse scan with A shifted left (:216-219);
    gA[t] = H[t-1]*gX[t], gA[0] = 0 (:221-222)."""
    L = A.shape[1]
    gX = torch.empty_like(gH)
    r = torch.zeros_like(gH[:, 0])
    for t in range(L - 1, -1, -1):
        r = gH[:, t] + (A[:, t + 1] * r if t + 1 < L else 0)
        gX[:, t] = r
    gA = torch.zeros_like(gH)
    gA[:, 1:] = H[:, :-1] * gX[:, 1:]
    return gA, gX


def selective_scan(x, delta, A, B, C, D):
    """MambaBlock.selective_scan_seq / selective_scan (cross_atten/mamba.py:265-318).
    x, delta: (B, L, ED); A: (ED, N); B, C: (B, L, N); D: (ED) -> (B, L, ED)."""
    deltaA = torch.exp(delta.unsqueeze(-1) * A)                      # mamba.py:275 / 300
    BX = delta.unsqueeze(-1) * B.unsqueeze(2) * x.unsqueeze(-1)      # mamba.py:276-278 / 301-303
    h = torch.zeros(x.shape[0], x.shape[2], A.shape[1], dtype=deltaA.dtype)
    ys = []
    # (unbind instead of [:, t]: the same values, but autograd's backward of a per-step slice allocates a full-size zero tensor per step)
    for a_t, bx_t, c_t in zip(deltaA.unbind(1), BX.unbind(1), C.unbind(1)):   # mamba.py:308-310
        h = a_t * h + bx_t
        ys.append((h * c_t.unsqueeze(1)).sum(-1))                    # mamba.py:314 (hs @ C)
    return torch.stack(ys, 1) + D * x                                # mamba.py:316


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False):
    """Contract of the reference's plug-in slot (cross_atten/mamba.py:243-252): channel-major layouts
    u, delta, z: (B, ED, L); B, C: (B, N, L).  Equivalent to the fallback path mamba.py:254-259 + 220-222."""
    d = delta.transpose(1, 2)
    if delta_bias is not None:
        d = d + delta_bias
    if delta_softplus:
        d = F.softplus(d)                                            # mamba.py:256
    x = u.transpose(1, 2)
    y = selective_scan(x, d, A, B.transpose(1, 2), C.transpose(1, 2),
                       D if D is not None else torch.zeros(A.shape[0], dtype=A.dtype))
    if z is not None:
        y = y * F.silu(z.transpose(1, 2))                            # mamba.py:220-222
    return y.transpose(1, 2)


def rmsnorm(x, w, eps=1e-5):
    """cross_atten/mamba.py:408-418."""
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def mamba_block(x, sd, pre, d_state=16, d_conv=4):
    """MambaBlock.forward + ssm (cross_atten/mamba.py:197-263), fallback (non-plug-in) arithmetic."""
    L = x.shape[1]
    ED = sd[pre + "in_proj.weight"].shape[0] // 2
    dt_rank = sd[pre + "dt_proj.weight"].shape[1]
    xz = x @ sd[pre + "in_proj.weight"].t()                          # :204 (bias=False, :48)
    xs, z = xz.chunk(2, dim=-1)                                      # :205
    xc = F.conv1d(xs.transpose(1, 2), sd[pre + "conv1d.weight"], sd[pre + "conv1d.bias"],
                  padding=d_conv - 1, groups=ED)[:, :, :L].transpose(1, 2)   # :208-210
    xc = F.silu(xc)                                                  # :212
    A = -torch.exp(sd[pre + "A_log"].float())                        # :232
    D = sd[pre + "D"].float()                                        # :233
    dbc = xc @ sd[pre + "x_proj.weight"].t()                         # :235
    delta, B, C = torch.split(dbc, [dt_rank, d_state, d_state], dim=-1)   # :236
    if (pre + "dt_layernorm.weight") in sd:                          # Jamba's inner_layernorms (:171-178, 188-195, 237)
        delta = rmsnorm(delta, sd[pre + "dt_layernorm.weight"])
        B = rmsnorm(B, sd[pre + "B_layernorm.weight"])
        C = rmsnorm(C, sd[pre + "C_layernorm.weight"])
    delta = (sd[pre + "dt_proj.weight"] @ delta.transpose(1, 2)).transpose(1, 2)   # :238, :255
    delta = F.softplus(delta + sd[pre + "dt_proj.bias"])             # :256
    y = selective_scan(xc, delta, A, B, C, D)                        # :259
    return (y * F.silu(z)) @ sd[pre + "out_proj.weight"].t()         # :220-223


def mamba(x, sd, pre, n_layers):
    """Mamba.forward / ResidualBlock.forward (cross_atten/mamba.py:69-77, 98-104)."""
    for l in range(n_layers):
        p = f"{pre}layers.{l}."
        x = mamba_block(rmsnorm(x, sd[p + "norm.weight"]), sd, p + "mixer.") + x
    return x


def jamba_mlp(x, sd, pre):
    """MLP (cross_atten/jamba.py:519-535): down(silu(gate(x)) * up(x)), no biases."""
    return (F.silu(x @ sd[pre + "gate_proj.weight"].t()) * (x @ sd[pre + "up_proj.weight"].t())) @ sd[pre + "down_proj.weight"].t()


def jamba_moe(x, sd, pre, top_k=2):
    """SparseMoEBlock.forward (cross_atten/jamba.py:459-517): one expert when there is no router; otherwise softmax over the router
    logits, top-k (NOT renormalised), every selected expert's output times its routing weight, summed per token."""
    B, L, D = x.shape
    if (pre + "router.weight") not in sd:
        return jamba_mlp(x, sd, pre + "experts.0.")                   # :470-478
    xf = x.reshape(-1, D)
    w = F.softmax(xf @ sd[pre + "router.weight"].t(), dim=1)         # :483-485
    w, sel = torch.topk(w, top_k, dim=-1)                            # :486
    out = torch.zeros_like(xf)
    n_exp = sd[pre + "router.weight"].shape[0]
    for e in range(n_exp):                                           # :496-513 (per expert: gather, MLP, scale, index_add)
        for j in range(top_k):
            m = sel[:, j] == e
            if m.any():
                out[m] = out[m] + jamba_mlp(xf[m], sd, f"{pre}experts.{e}.") * w[m, j, None]
    return out.reshape(B, L, D)


def jamba_attention(x, sd, pre, n_heads, n_kv_heads):
    """AttentionSDPA.forward without a cache (cross_atten/jamba.py:362-398): q/k/v projections (no bias), GQA repeat, causal
    scaled-dot-product attention, output projection."""
    B, L, D = x.shape
    dh = D // n_heads
    q = (x @ sd[pre + "q_proj.weight"].t()).view(B, L, n_heads, dh).transpose(1, 2)
    k = (x @ sd[pre + "k_proj.weight"].t()).view(B, L, n_kv_heads, dh).transpose(1, 2)
    v = (x @ sd[pre + "v_proj.weight"].t()).view(B, L, n_kv_heads, dh).transpose(1, 2)
    rep = n_heads // n_kv_heads
    if rep > 1:                                                      # repeat_kv (:558-567)
        k, v = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    s = s.masked_fill(torch.ones(L, L, dtype=torch.bool).triu(1), float("-inf"))      # is_causal=True (:390-392)
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, D)
    return o @ sd[pre + "o_proj.weight"].t()


def jamba(x, sd, pre, n_layers, n_heads, n_kv_heads=8, attn_offset=4, attn_period=8):
    """Jamba.forward (cross_atten/jamba.py:258-296) with AttentionLayer / MambaLayer (:308-340, 400-439): pre-norm residual mixer, then
    pre-norm residual MoE; attention at layers (i - 4) % 8 == 0, Mamba (with inner layernorms) elsewhere."""
    for i in range(n_layers):
        p = f"{pre}layers.{i}."
        h = rmsnorm(x, sd[p + "input_layernorm.weight"])
        if (i - attn_offset) % attn_period == 0:
            x = x + jamba_attention(h, sd, p + "self_attn.", n_heads, n_kv_heads)
        else:
            x = x + mamba_block(h, sd, p + "mamba.")
        x = x + jamba_moe(rmsnorm(x, sd[p + "pre_moe_layernorm.weight"]), sd, p + "moe.")
    return x


# ------------------------------------------------------------------------------------------------
# Group B -- tabular tokeniser, cross attention, head
# ------------------------------------------------------------------------------------------------


def categories_offset(categories, num_special_tokens=2):
    """cross_atten/mamba_transformer.py:44-46."""
    off = F.pad(torch.tensor(list(categories)), (1, 0), value=num_special_tokens)
    return off.cumsum(dim=-1)[:-1]


def build_condition(image_condition):
    """cross_atten/mamba_transformer.py:89-94: (B,1,D1,D2,D3) x2 -> (B, 2*D3, D1*D2)."""
    outs = []
    for img in image_condition:
        b, c, h, w, d = img.shape
        outs.append(img.reshape(b * c, h * w, d).transpose(1, 2).contiguous())
    return torch.cat(outs, dim=1)


def cross_attention(x, y, sd, pre, n_heads):
    """CrossAttention.forward (cross_atten/sd_cross_atten.py:49-70)."""
    Bsz, _, E = x.shape
    dh = E // n_heads
    lin = lambda t, n: t @ sd[pre + n + ".weight"].t() + sd[pre + n + ".bias"]
    q = lin(x, "q_proj").view(Bsz, -1, n_heads, dh).transpose(1, 2)
    k = lin(y, "k_proj").view(Bsz, -1, n_heads, dh).transpose(1, 2)
    v = lin(y, "v_proj").view(Bsz, -1, n_heads, dh).transpose(1, 2)
    w = (q @ k.transpose(-1, -2)) / math.sqrt(dh)                    # :61-62
    w = F.softmax(w, dim=-1)                                         # :63
    o = (w @ v).transpose(1, 2).contiguous().view(x.shape)           # :65-67
    return lin(o, "out_proj")                                        # :68


def geglu_ff(x, sd, pre, drop_mask=None):
    """FeedForward/GEGLU (cross_atten/corss_ft_transformer.py:10-22); drop_mask: pre-scaled dropout mask or None (eval)."""
    h = F.layer_norm(x, x.shape[-1:], sd[pre + "0.weight"], sd[pre + "0.bias"])
    h = h @ sd[pre + "1.weight"].t() + sd[pre + "1.bias"]
    a, g = h.chunk(2, dim=-1)
    h = a * F.gelu(g)
    if drop_mask is not None:
        h = h * drop_mask
    return h @ sd[pre + "4.weight"].t() + sd[pre + "4.bias"]


def cross_mamba_both(x_categ, x_numer, feature_img, image_condition, sd, depth, heads, pre="", drop_mask=None):
    """Cross_mamba_both.forward (cross_atten/mamba_transformer.py:87-133)."""
    cond = build_condition(image_condition)                          # :89-94
    xs = []
    if (pre + "categorical_embeds.weight") in sd:
        idx = x_categ + sd[pre + "categories_offset"]                # :98
        xs.append(sd[pre + "categorical_embeds.weight"][idx])        # :100
    if (pre + "numerical_embedder.weights") in sd:                   # corss_ft_transformer.py:159-163
        xs.append(x_numer.unsqueeze(-1) * sd[pre + "numerical_embedder.weights"] + sd[pre + "numerical_embedder.biases"])
    x = torch.cat(xs, dim=1)                                         # :112
    cls = sd[pre + "cls_token"].expand(x.shape[0], -1, -1)           # :116
    x = torch.cat((cls, x, feature_img), dim=1)                      # :117
    x = mamba(x, sd, pre + "transformer.", depth)                    # :121
    x = x.mean(dim=1, keepdim=True)                                  # :122
    x = cross_attention(x, cond, sd, pre + "final_cross.", heads) + x   # :124
    x = geglu_ff(x, sd, pre + "final_feed.", drop_mask) + x          # :125
    x = x.squeeze(1)                                                 # :127
    x = F.layer_norm(x, x.shape[-1:], sd[pre + "to_logits.0.weight"], sd[pre + "to_logits.0.bias"])
    return x @ sd[pre + "to_logits.1.weight"].t() + sd[pre + "to_logits.1.bias"]   # :131


def cross_jamba_both(x_categ, x_numer, feature_img, image_condition, sd, depth, heads, pre="", drop_mask=None):
    """Cross_jamba_both.forward (cross_atten/mamba_transformer.py:203-251): Cross_mamba_both with the Jamba backbone (2*depth layers)."""
    cond = build_condition(image_condition)                          # :205-210
    xs = []
    if (pre + "categorical_embeds.weight") in sd:
        xs.append(sd[pre + "categorical_embeds.weight"][x_categ + sd[pre + "categories_offset"]])      # :214-218
    if (pre + "numerical_embedder.weights") in sd:
        xs.append(x_numer.unsqueeze(-1) * sd[pre + "numerical_embedder.weights"] + sd[pre + "numerical_embedder.biases"])
    x = torch.cat(xs, dim=1)
    cls = sd[pre + "cls_token"].expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x, feature_img), dim=1)                      # :235
    x = jamba(x, sd, pre + "transformer.", 2 * depth, heads)         # :239 (num_key_value_heads keeps its default 8)
    x = x.mean(dim=1, keepdim=True)                                  # :240
    x = cross_attention(x, cond, sd, pre + "final_cross.", heads) + x   # :242
    x = geglu_ff(x, sd, pre + "final_feed.", drop_mask) + x          # :243
    x = x.squeeze(1)
    x = F.layer_norm(x, x.shape[-1:], sd[pre + "to_logits.0.weight"], sd[pre + "to_logits.0.bias"])
    return x @ sd[pre + "to_logits.1.weight"].t() + sd[pre + "to_logits.1.bias"]   # :249


def cross_mamba_ablation(x_categ, x_numer, feature_img, image_condition, sd, depth, heads, pre="", no_table=False, drop_mask=None):
    """Cross_mamba_ablation.forward (cross_atten/mamba_transformer.py:327-385): Cross_mamba_both with three switches --
    feature_img=None (table only), no_table=True (cls + image features only), image_condition=None (no cross attention)."""
    xs = []
    if (pre + "categorical_embeds.weight") in sd:
        xs.append(sd[pre + "categorical_embeds.weight"][x_categ + sd[pre + "categories_offset"]])      # :339-343
    if (pre + "numerical_embedder.weights") in sd:
        xs.append(x_numer.unsqueeze(-1) * sd[pre + "numerical_embedder.weights"] + sd[pre + "numerical_embedder.biases"])
    x = torch.cat(xs, dim=1)                                         # :353
    cls = sd[pre + "cls_token"].expand(x.shape[0], -1, -1)
    if no_table:
        x = torch.cat((cls, feature_img), dim=1)                     # :359
    elif feature_img is not None:
        x = torch.cat((cls, x, feature_img), dim=1)                  # :362
    else:
        x = torch.cat((cls, x), dim=1)                               # :364
    x = mamba(x, sd, pre + "transformer.", depth)                    # :370
    x = x.mean(dim=1, keepdim=True)                                  # :371
    if image_condition is not None:                                  # :373-375
        cond = build_condition(image_condition)
        x = cross_attention(x, cond, sd, pre + "final_cross.", heads) + x
        x = geglu_ff(x, sd, pre + "final_feed.", drop_mask) + x
    x = x.squeeze(1)
    x = F.layer_norm(x, x.shape[-1:], sd[pre + "to_logits.0.weight"], sd[pre + "to_logits.0.bias"])
    return x @ sd[pre + "to_logits.1.weight"].t() + sd[pre + "to_logits.1.bias"]   # :381


def combine_classifier_vit_mid(mid_input, mid_output, sd, pre=""):
    """Combine_classfier_vit_mid.forward (classify/classifier.py:329-333)."""
    t = torch.cat([mid_input, mid_output], dim=1).flatten(2)        # b c (h w)
    f = t @ sd[pre + "vit_mid_linear.weight"].t() + sd[pre + "vit_mid_linear.bias"]
    return f.transpose(1, 2).contiguous()


# ------------------------------------------------------------------------------------------------
# Group C -- frozen generator (Residual_mid_UNet3D_vit, eval, forward only)
# ------------------------------------------------------------------------------------------------


_PATTERN = None          # tests only: recorded ReLU / max-pool selections replayed in call order (activation_pattern)


class activation_pattern:
    """Replay a recorded activation pattern inside generator(): every ReLU multiplies by the next recorded 0/1 mask and every 2x2x2
    max-pool takes the recorded one-hot selection, in call order.  The gradient of a piecewise-linear network is a function of its
    activation pattern; a bf16 forward flips ~0.1 % of near-zero ReLUs against fp32, each flip an O(1) change of some gradient terms,
    so gradient parity is checked at the SAME pattern (tests/test_gen_train_gpu.py) and the flipped fraction is checked separately."""

    def __init__(self, masks):
        self.masks = list(masks)

    def __enter__(self):
        global _PATTERN
        _PATTERN = iter(self.masks)
        return self

    def __exit__(self, *exc):
        global _PATTERN
        _PATTERN = None


def _relu(x):
    return F.relu(x) if _PATTERN is None else x * next(_PATTERN).to(x.dtype)


def _max_pool2(x):
    if _PATTERN is None:
        return F.max_pool3d(x, 2)
    return F.avg_pool3d(x * next(_PATTERN).to(x.dtype), 2) * 8.0      # the one-hot selection summed over each window


def single_conv(x, sd, pre, relu, num_groups=8):
    """SingleConv order 'gcr'/'gc' (pytorch3dunet/unet3d/buildingblocks.py:38-67, 108-115): GroupNorm over
    in_channels -> Conv3d k3 p1 no bias -> optional ReLU."""
    C = x.shape[1]
    g = num_groups if C >= num_groups else 1                         # :62-63
    x = F.group_norm(x, g, sd[pre + "groupnorm.weight"], sd[pre + "groupnorm.bias"], eps=1e-5)
    x = F.conv3d(x, sd[pre + "conv.weight"], None, padding=1)
    return _relu(x) if relu else x


def resnet_block(x, sd, pre):
    """ResNetBlock.forward, order 'gcr' => ReLU (buildingblocks.py:211-229)."""
    if (pre + "conv1.weight") in sd:
        r = F.conv3d(x, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"])   # :191-196 (1x1x1, bias)
    else:
        r = x                                                        # :198 Identity
    o = single_conv(r, sd, pre + "conv2.", relu=True)
    o = single_conv(o, sd, pre + "conv3.", relu=False)
    return _relu(o + r)                                              # :226-227


def fold_mid(x, md1=8):
    """'b c (md1 md2) h w -> b c (h md1) (md2 w)' (pytorch3dunet/unet3d/model.py:150)."""
    b, c, d, h, w = x.shape
    md2 = d // md1
    return x.view(b, c, md1, md2, h, w).permute(0, 1, 4, 2, 3, 5).reshape(b, c, h * md1, md2 * w)


def unfold_mid(y, w, md1=8):
    """'b c (h md1) (md2 w) -> b c (md1 md2) h w' (model.py:152)."""
    b, c, H, W = y.shape
    h, md2 = H // md1, W // w
    return y.view(b, c, h, md1, md2, w).permute(0, 1, 3, 4, 2, 5).reshape(b, c, md1 * md2, h, w)


def patchify(img, p):
    """'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (vit_pytorch_diy/vit.py:96)."""
    b, c, H, W = img.shape
    h, w = H // p, W // p
    return img.view(b, c, h, p, w, p).permute(0, 2, 4, 3, 5, 1).reshape(b, h * w, p * p * c)


def unpatchify(t, p, h, c):
    """'b (h w) (p1 p2 c) -> b c (h p1) (w p2)' (vit.py:109)."""
    b, n, _ = t.shape
    w = n // h
    return t.view(b, h, w, p, p, c).permute(0, 5, 1, 3, 2, 4).reshape(b, c, h * p, w * p)


def _ln(x, sd, pre):
    return F.layer_norm(x, x.shape[-1:], sd[pre + "weight"], sd[pre + "bias"], eps=1e-5)


def _linear(x, sd, pre):
    y = x @ sd[pre + "weight"].t()
    return y + sd[pre + "bias"] if (pre + "bias") in sd else y


def vit_attention(x, sd, pre, heads):
    """Attention.forward (vit.py:50-63), eval mode."""
    b, n, _ = x.shape
    h = _ln(x, sd, pre + "norm.")
    qkv = h @ sd[pre + "to_qkv.weight"].t()
    inner = qkv.shape[-1] // 3
    dh = inner // heads
    q, k, v = [t.view(b, n, heads, dh).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    dots = (q @ k.transpose(-1, -2)) * dh ** -0.5
    out = (F.softmax(dots, dim=-1) @ v).transpose(1, 2).reshape(b, n, inner)
    return _linear(out, sd, pre + "to_out.0.")


def vit_feedforward(x, sd, pre):
    """FeedForward (vit.py:14-27), eval mode, exact-erf GELU."""
    h = _ln(x, sd, pre + "net.0.")
    h = F.gelu(_linear(h, sd, pre + "net.1."))
    return _linear(h, sd, pre + "net.4.")


def vit_mid(img, sd, pre, patch, heads, depth):
    """ViT.forward incl. from_patch_embedding (vit.py:95-110, 124-137), eval mode."""
    b, c, H, W = img.shape
    x = patchify(img, patch)
    x = _ln(x, sd, pre + "to_patch_embedding.1.")
    x = _linear(x, sd, pre + "to_patch_embedding.2.")
    x = _ln(x, sd, pre + "to_patch_embedding.3.")
    n = x.shape[1]
    x = torch.cat((sd[pre + "cls_token"].expand(b, -1, -1), x), dim=1)   # :127-128
    x = x + sd[pre + "pos_embedding"][:, :n + 1]                     # :130
    for l in range(depth):                                           # vit.py:76-79
        x = vit_attention(x, sd, f"{pre}transformer.layers.{l}.0.", heads) + x
        x = vit_feedforward(x, sd, f"{pre}transformer.layers.{l}.1.") + x
    x = _ln(x, sd, pre + "transformer.norm.")                        # :81
    x = _ln(x, sd, pre + "from_patch_embedding.0.")                  # :103
    x = _linear(x.transpose(1, 2), sd, pre + "from_patch_embedding.2.").transpose(1, 2)   # :104-106 (token axis)
    x = _linear(x, sd, pre + "from_patch_embedding.4.")              # :107
    x = _ln(x, sd, pre + "from_patch_embedding.5.")                  # :108
    return unpatchify(x, patch, H // patch, c)                       # :109


def patchify3d(video, pf, p1, p2):
    """'b c (f pf) (h p1) (w p2) -> b (f h w) (p1 p2 pf c)' (vit_pytorch_diy/vit_3d.py:93)."""
    b, c, F_, H, W = video.shape
    f, h, w = F_ // pf, H // p1, W // p2
    return video.view(b, c, f, pf, h, p1, w, p2).permute(0, 2, 4, 6, 5, 7, 3, 1).reshape(b, f * h * w, p1 * p2 * pf * c)


def vit3d(video, sd, pre, frame_patch, patch, heads, depth, pool="cls"):
    """vit_3d.ViT.forward (vit_3d.py:113-128), eval mode: no final transformer norm, cls / mean pool, mlp_head."""
    b = video.shape[0]
    x = patchify3d(video, frame_patch, patch, patch)
    x = _ln(x, sd, pre + "to_patch_embedding.1.")
    x = _linear(x, sd, pre + "to_patch_embedding.2.")
    x = _ln(x, sd, pre + "to_patch_embedding.3.")
    tokens = x
    n = x.shape[1]
    x = torch.cat((sd[pre + "cls_token"].expand(b, -1, -1), x), dim=1)
    x = x + sd[pre + "pos_embedding"][:, :n + 1]
    for l in range(depth):
        x = vit_attention(x, sd, f"{pre}transformer.layers.{l}.0.", heads) + x
        x = vit_feedforward(x, sd, f"{pre}transformer.layers.{l}.1.") + x
    x = x.mean(dim=1) if pool == "mean" else x[:, 0]
    return _linear(_ln(x, sd, pre + "mlp_head.0."), sd, pre + "mlp_head.1."), tokens


def nearest_resize(x, size):
    """F.interpolate(x, size) default nearest (buildingblocks.py:523-531): dst i <- src floor(i*in/out)."""
    for dim, s in zip((2, 3, 4), size):
        n = x.shape[dim]
        if n != s:
            idx = torch.div(torch.arange(s) * n, s, rounding_mode="floor")
            x = x.index_select(dim, idx)
    return x


def decoder(enc_feat, x, sd, pre):
    """Decoder.forward for ResNetBlock: ConvTranspose3d k3 s2 p1 no bias -> nearest resize -> sum join -> ResNetBlock
    (buildingblocks.py:389-400, 523-537)."""
    x = F.conv_transpose3d(x, sd[pre + "upsampling.upsample.conv_transposed.weight"], None, stride=2, padding=1)
    x = nearest_resize(x, enc_feat.shape[2:])
    return resnet_block(enc_feat + x, sd, pre + "basic_module.")


def generator(x, sd, pre="", levels=3, vit_patch=None, vit_heads=6, vit_depth=4, md1=8):
    """Mid_UNet_vit.forward(x, output_vit_mid=True) in eval mode (pytorch3dunet/unet3d/model.py:137-175).
    Returns (mid_input, mid_output, pet)."""
    feats = []
    for i in range(levels):
        if i > 0:
            x = _max_pool2(x)                                        # buildingblocks.py:284, 306-307
        x = resnet_block(x, sd, f"{pre}encoders.{i}.basic_module.")
        feats.insert(0, x)
    feats = feats[1:]
    mid_input = fold_mid(x, md1)                                     # model.py:150
    if vit_patch is None:
        vit_patch = x.shape[3]                                       # patch = D2/4 (SURVEY 8-d geometry rule)
    mid_output = vit_mid(mid_input, sd, pre + "mid.", vit_patch, vit_heads, vit_depth)
    x = unfold_mid(mid_output, x.shape[-1], md1)                     # model.py:152
    for j, ef in enumerate(feats):
        x = decoder(ef, x, sd, f"{pre}decoders.{j}.")
    pet = F.conv3d(x, sd[pre + "final_conv.weight"], sd[pre + "final_conv.bias"])   # model.py:162
    return mid_input, mid_output, pet


# ------------------------------------------------------------------------------------------------
# Group G -- the step
# ------------------------------------------------------------------------------------------------


def bce_sigmoid(pred, y):
    """classify_mamba.py:104: BCELoss(sigmoid(pred.squeeze(1)), y.float()), mean."""
    return F.binary_cross_entropy(torch.sigmoid(pred.squeeze(1)), y.float())


def clip_per_param(grads, max_norm=1.0):
    """classify_mamba.py:106-107: clip_grad_norm_ applied to each parameter on its own."""
    out = []
    for g in grads:
        coef = max_norm / (g.norm(2) + 1e-6)
        out.append(g * torch.clamp(coef, max=1.0))
    return out


def adam_step(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (classify_mamba.py:64)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat = m / (1 - b1 ** step)
    vhat = v / (1 - b2 ** step)
    return p - lr * mhat / (vhat.sqrt() + eps), m, v


def geometry(D1, D2, D3, md1=8):
    """Derived constants for non-native volume sizes: returns dict(H, W, patch, num_patches, d_cross, keys)."""
    assert D1 % 32 == 0 and D2 % 4 == 0 and D3 % 4 == 0
    H, W, p = (D2 // 4) * md1, (D1 // 32) * (D3 // 4), D2 // 4
    assert W % p == 0
    return dict(H=H, W=W, patch=p, num_patches=(H // p) * (W // p), d_cross=D1 * D2, keys=2 * D3)


# ------------------------------------------------------------------------------------------------
# Input pipeline
# ------------------------------------------------------------------------------------------------


def adaptive_normal(img, min_p=0.001, max_p=0.999):
    """utils/data_normalization.py:20-48: quantiles of the voxels >= 0 by a full sort, affine map to [-1, 1], clip."""
    pix = torch.sort(img[img >= 0])[0]                                # :26-27
    n = pix.numel()

    def at(p):                                                         # :28-33 / :35-40
        idx = int(round(n - 1) * p + 0.5)
        return pix[min(max(idx, 0), n - 1)]
    lo, hi = at(min_p), at(max_p)
    mean, sd = (hi + lo) / 2.0, (hi - lo) / 2.0                       # :42-43
    out = (img - mean) / sd                                           # :44
    out[out < -1] = -1.0                                              # :45
    out[out > 1] = 1.0                                                # :46
    return out
