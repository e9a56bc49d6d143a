"""lucidrains 3-D ViT (reference: vit_pytorch_diy/vit_3d.py:78-128) -- MI355X build.

Same constructor kwargs and state-dict keys (to_patch_embedding.{1,2,3}, pos_embedding, cls_token,
transformer.layers.{l}.{0,1}..., mlp_head.{0,1}).  The reference never instantiates this module on its hot path; SURVEY 8-d
uses it as the MFMA-attention measurement row (image 96, patch 8, frames 96 -> 1729 tokens).  torch layers only hold
parameters; forward() runs the HIP kernels: LayerNorm, bf16 MFMA GEMM (bias / GELU / residual epilogues) and the flash-style
attention kernel gfe_attention_fwd.  Under no_grad: the inference pipeline (bf16 activations, fused epilogues).  With autograd enabled
and trainable parameters: forward_train -- the same module through differentiable kernels (flash-attention forward with the row
statistic + gfe_attention_bwd, LayerNorm and Linear nodes of the classifier head), which is how the reference trains it.
"""
import torch
from torch import nn

from gfe_hip import nn_ops as K
from gfe_hip.nn_ops import BF16
from vit_pytorch_diy.vit import FeedForward, Attention, pair


class Transformer(nn.Module):          # vit_3d.py:60-74 (no final norm, unlike vit.py)
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout),
                                                    FeedForward(dim, mlp_dim, dropout=dropout)]) for _ in range(depth)])


class ViT(nn.Module):
    def __init__(self, *, image_size, image_patch_size, frames, frame_patch_size, num_classes, dim, depth, heads, mlp_dim,
                 pool='cls', channels=3, dim_head=64, dropout=0., emb_dropout=0.):
        super().__init__()
        image_height, image_width = pair(image_size)
        patch_height, patch_width = pair(image_patch_size)
        assert image_height % patch_height == 0 and image_width % patch_width == 0, 'Image dimensions must be divisible by the patch size.'
        assert frames % frame_patch_size == 0, 'Frames must be divisible by frame patch size'
        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'
        num_patches = (image_height // patch_height) * (image_width // patch_width) * (frames // frame_patch_size)
        patch_dim = channels * patch_height * patch_width * frame_patch_size
        self.geom = (frames, image_height, image_width, frame_patch_size, patch_height, patch_width)
        self.channels, self.dim, self.num_patches, self.patch_dim, self.num_classes = channels, dim, num_patches, patch_dim, num_classes
        # index 0 is the reference's parameter-free Rearrange (vit_3d.py:93)
        self.to_patch_embedding = nn.Sequential(nn.Identity(), nn.LayerNorm(patch_dim), nn.Linear(patch_dim, dim), nn.LayerNorm(dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))
        self._sig, self._w = None, None

    def _weights(self):
        params = list(self.parameters())
        sig = tuple((p.data_ptr(), p._version, str(p.device), getattr(p, "_gfe_epoch", (0,))[0]) for p in params)   # FlatAdam rewrites storage without a version bump
        if sig != self._sig:
            with torch.no_grad():
                f = lambda p: p.detach().float().contiguous()
                h = lambda p: K.cast(p.detach().float(), BF16)
                tpe = self.to_patch_embedding
                nc4 = -(-self.num_classes // 4) * 4                      # the GEMM stores 4 columns per lane
                wh = torch.zeros((nc4, self.dim), dtype=torch.float32, device=self.cls_token.device)
                wh[:self.num_classes] = f(self.mlp_head[1].weight)
                bh = torch.zeros(nc4, dtype=torch.float32, device=wh.device)
                bh[:self.num_classes] = f(self.mlp_head[1].bias)
                w = dict(ln_p=(f(tpe[1].weight), f(tpe[1].bias)), w_embed=h(tpe[2].weight), b_embed=f(tpe[2].bias),
                         ln_e=(f(tpe[3].weight), f(tpe[3].bias)), cls=f(self.cls_token).view(-1), pos=f(self.pos_embedding).view(-1, self.dim),
                         ln_h=(f(self.mlp_head[0].weight), f(self.mlp_head[0].bias)), w_head=K.cast(wh, BF16), b_head=bh, layers=[])
                for attn, ff in self.transformer.layers:
                    w["layers"].append(dict(ln1=(f(attn.norm.weight), f(attn.norm.bias)), wqkv=h(attn.to_qkv.weight),
                                            wo=h(attn.to_out[0].weight), bo=f(attn.to_out[0].bias),
                                            ln2=(f(ff.net[0].weight), f(ff.net[0].bias)), w1=h(ff.net[1].weight), b1=f(ff.net[1].bias),
                                            w2=h(ff.net[4].weight), b2=f(ff.net[4].bias)))
            self._w, self._sig = w, sig
        return self._w

    def patchify(self, video):
        """'b c (f pf) (h p1) (w p2) -> b (f h w) (p1 p2 pf c)' (vit_3d.py:93): pure indexing, one strided copy."""
        F_, H, W, pf, p1, p2 = self.geom
        B, C = video.shape[:2]
        v = video.view(B, C, F_ // pf, pf, H // p1, p1, W // p2, p2).permute(0, 2, 4, 6, 5, 7, 3, 1)
        return v.reshape(B * self.num_patches, self.patch_dim)

    def tokens(self, video):
        """to_patch_embedding output (B, n, dim) f32 (vit_3d.py:92-97)."""
        w, n, dim = self._weights(), self.num_patches, self.dim
        B = video.shape[0]
        rows = self.patchify(video.float()).contiguous()
        tok = K.layernorm(rows, *w["ln_p"], rows=B * n, length=self.patch_dim, out_dtype=BF16)
        emb = K.gemm_nt(tok, w["w_embed"], bias=w["b_embed"], out_dtype=torch.float32)
        return K.layernorm(emb, *w["ln_e"], rows=B * n, length=dim, out_dtype=torch.float32).view(B, n, dim)

    def forward_train(self, video):
        """vit_3d.py:113-128 with autograd (module.training decides the dropouts): every op is a HIP-kernel autograd node; attention is
        train_ops.qkv_flash_attention for dim_head 64 (any token count), the 64-token kernel of the classifier head otherwise."""
        from gfe_hip.head_ops import dropout, gelu, layernorm_rows, sdpa_small
        from gfe_hip.train_ops import linear as linear_, qkv_flash_attention
        linear = lambda a, w, b: linear_(a, w, b, exact=False)         # bf16 MFMA operands, like the inference pipeline and the attention
        n, dim = self.num_patches, self.dim
        B, T = video.shape[0], n + 1
        drop = lambda x, m: dropout(x, m.p, m.training)                    # nn.Dropout on the residual branches: hash mask, regenerated in the backward
        tpe = self.to_patch_embedding
        t = self.patchify(video.float()).contiguous()
        t = layernorm_rows(t, tpe[1].weight, tpe[1].bias, tpe[1].eps)
        t = linear(t, tpe[2].weight, tpe[2].bias)
        t = layernorm_rows(t, tpe[3].weight, tpe[3].bias, tpe[3].eps).view(B, n, dim)
        x = torch.cat((self.cls_token.expand(B, -1, -1), t), dim=1) + self.pos_embedding[:, :T]       # vit_3d.py:117-119
        x = drop(x, self.dropout)
        for attn, ff in self.transformer.layers:                                                      # vit_3d.py:70-73
            h = layernorm_rows(x, attn.norm.weight, attn.norm.bias, attn.norm.eps)
            p_attn = attn.dropout.p if (attn.training and attn.dropout.training) else 0.0
            if attn.dim_head == 64:
                o = qkv_flash_attention(h, attn.to_qkv.weight, attn.heads, attn.scale, dropout_p=p_attn)
            else:
                if T > 64 or attn.dim_head > 64:
                    raise NotImplementedError("vit_3d training: dim_head != 64 is built for <= 64 tokens only")
                q, k, v = linear(h, attn.to_qkv.weight, None).chunk(3, dim=-1)
                o = sdpa_small(q.contiguous(), k.contiguous(), v.contiguous(), attn.heads, causal=False, dropout_p=p_attn)
            x = drop(linear(o, attn.to_out[0].weight, attn.to_out[0].bias), attn.to_out[1]) + x
            h = layernorm_rows(x, ff.net[0].weight, ff.net[0].bias, ff.net[0].eps)
            h = drop(gelu(linear(h, ff.net[1].weight, ff.net[1].bias)), ff.net[3])                 # exact-erf GELU, one kernel each way
            x = drop(linear(h, ff.net[4].weight, ff.net[4].bias), ff.net[5]) + x
        pooled = x.mean(dim=1) if self.pool == 'mean' else x[:, 0]                                    # vit_3d.py:125
        hn = layernorm_rows(pooled.contiguous(), self.mlp_head[0].weight, self.mlp_head[0].bias, self.mlp_head[0].eps)
        return linear_(hn, self.mlp_head[1].weight, self.mlp_head[1].bias)                           # (B x dim: exact f32)

    def forward(self, video):
        """video: (B, C, F, H, W) float on the GPU -> (B, num_classes) f32.  vit_3d.py:113-128."""
        assert video.is_cuda, "no CPU fallback"
        F_, H, W = self.geom[:3]
        assert tuple(video.shape[1:]) == (self.channels, F_, H, W)
        if torch.is_grad_enabled() and (video.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self.forward_train(video)
        w, n, dim = self._weights(), self.num_patches, self.dim
        B, T = video.shape[0], n + 1
        emb = self.tokens(video).view(B * n, dim)
        x = K.vit_embed(emb, w["cls"], w["pos"], B, n, dim).view(B * T, dim)                  # cls + pos (vit_3d.py:117-119)
        a0 = self.transformer.layers[0][0]
        heads, dh = a0.heads, a0.dim_head
        inner = heads * dh
        for lw in w["layers"]:                                                              # vit_3d.py:70-73
            h = K.layernorm(x, *lw["ln1"], rows=B * T, length=dim, out_dtype=BF16)
            qkv = K.gemm_nt(h, lw["wqkv"])
            if dh == 64:
                o = K.attention_fwd(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], B, heads, T, dh, a0.scale)
            else:
                assert T <= 256 and dh <= 64, "attention shape outside the built kernels (dim_head 64, or <= 256 tokens)"
                o = K.attention_small(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], B, heads, T, T, dh, a0.scale)
            x = K.gemm_nt(o, lw["wo"], bias=lw["bo"], res=x, out_dtype=torch.float32)
            h = K.layernorm(x, *lw["ln2"], rows=B * T, length=dim, out_dtype=BF16)
            h = K.gemm_nt(h, lw["w1"], bias=lw["b1"], act=1)
            x = K.gemm_nt(h, lw["w2"], bias=lw["b2"], res=x, out_dtype=torch.float32)
        x = x.view(B, T, dim)
        pooled = (x.mean(dim=1) if self.pool == 'mean' else x[:, 0]).contiguous()           # vit_3d.py:125
        hn = K.layernorm(pooled, *w["ln_h"], rows=B, length=dim, out_dtype=BF16)
        return K.gemm_nt(hn, w["w_head"], bias=w["b_head"], out_dtype=torch.float32)[:, :self.num_classes]
