"""Mamba stack -- MI355X build of the reference's cross_atten/mamba.py (MambaConfig :31-59, Mamba :61-89,
ResidualBlock :91-117, MambaBlock :119-405, RMSNorm :408-418).  Same class names, constructor arguments, attributes
and state-dict keys (in_proj, conv1d, x_proj, dt_proj, A_log, D, out_proj, norm).

MambaBlock.forward always takes the fused route the reference only reaches through its optional `selective_scan_fn`
plug-in (mamba.py:243-252): softplus(delta + dt_proj.bias), the scan, D*x and the y*silu(z) gate run in one HIP kernel
(gfe-mamba_amd/csrc/sscan.hip) in the token-major layout, so the four transposes of mamba.py:245-252 disappear.
Projections run on the bf16 MFMA GEMM.  `config.pscan` / `config.use_cuda` are accepted and ignored (one path).
"""
import math
import os
from dataclasses import dataclass
from typing import Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from gfe_hip.scan_ops import selective_scan_fn, selective_scan_tm
from gfe_hip import mamba_block as _fused
from gfe_hip.train_ops import Linear, dwconv1d_silu, rmsnorm


@dataclass
class MambaConfig:
    d_model: int
    n_layers: int
    dt_rank: Union[int, str] = 'auto'
    d_state: int = 16
    expand_factor: int = 2
    d_conv: int = 4
    dt_min: float = 0.001
    dt_max: float = 0.1
    dt_init: str = "random"
    dt_scale: float = 1.0
    dt_init_floor = 1e-4
    rms_norm_eps: float = 1e-5
    bias: bool = False
    conv_bias: bool = True
    inner_layernorms: bool = False
    pscan: bool = True
    use_cuda: bool = False

    def __post_init__(self):
        self.d_inner = self.expand_factor * self.d_model
        if self.dt_rank == 'auto':
            self.dt_rank = math.ceil(self.d_model / 16)


class Mamba(nn.Module):
    def __init__(self, config: MambaConfig):
        super().__init__()
        self.config = config
        self.layers = nn.ModuleList([ResidualBlock(config) for _ in range(config.n_layers)])

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x

    def step(self, x, caches):
        for i, layer in enumerate(self.layers):
            x, caches[i] = layer.step(x, caches[i])
        return x, caches


class ResidualBlock(nn.Module):
    def __init__(self, config: MambaConfig):
        super().__init__()
        self.mixer = MambaBlock(config)
        self.norm = RMSNorm(config.d_model, config.rms_norm_eps)

    def forward(self, x):
        return self.mixer(self.norm(x)) + x                      # mamba.py:103

    def step(self, x, cache):
        output, cache = self.mixer.step(self.norm(x), cache)
        return output + x, cache


class MambaBlock(nn.Module):
    def __init__(self, config: MambaConfig):
        super().__init__()
        self.config = config
        self.in_proj = Linear(config.d_model, 2 * config.d_inner, bias=config.bias)
        self.conv1d = nn.Conv1d(in_channels=config.d_inner, out_channels=config.d_inner, kernel_size=config.d_conv,
                                bias=config.conv_bias, groups=config.d_inner, padding=config.d_conv - 1)
        self.x_proj = Linear(config.d_inner, config.dt_rank + 2 * config.d_state, bias=False)
        self.dt_proj = Linear(config.dt_rank, config.d_inner, bias=True)
        dt_init_std = config.dt_rank ** -0.5 * config.dt_scale   # mamba.py:141-147
        if config.dt_init == "constant":
            nn.init.constant_(self.dt_proj.weight, dt_init_std)
        elif config.dt_init == "random":
            nn.init.uniform_(self.dt_proj.weight, -dt_init_std, dt_init_std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(config.d_inner) * (math.log(config.dt_max) - math.log(config.dt_min))
                       + math.log(config.dt_min)).clamp(min=config.dt_init_floor)   # mamba.py:150-155
        inv_dt = dt + torch.log(-torch.expm1(-dt))
        with torch.no_grad():
            self.dt_proj.bias.copy_(inv_dt)
        A = torch.arange(1, config.d_state + 1, dtype=torch.float32).repeat(config.d_inner, 1)
        self.A_log = nn.Parameter(torch.log(A))                  # mamba.py:160-162
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(config.d_inner))        # mamba.py:164-165
        self.D._no_weight_decay = True
        self.out_proj = Linear(config.d_inner, config.d_model, bias=config.bias)
        if config.inner_layernorms:                              # used in Jamba (mamba.py:171-178)
            self.dt_layernorm = RMSNorm(config.dt_rank, config.rms_norm_eps)
            self.B_layernorm = RMSNorm(config.d_state, config.rms_norm_eps)
            self.C_layernorm = RMSNorm(config.d_state, config.rms_norm_eps)
        else:
            self.dt_layernorm = self.B_layernorm = self.C_layernorm = None
        # the reference's plug-in slot (mamba.py:180-186); here it is always populated, never a silent fallback
        self.selective_scan_cuda = selective_scan_fn

    def _conv_silu(self, x):
        """Depthwise causal conv1d (k = d_conv, left padding) + bias + SiLU on (B, L, ED)  (mamba.py:208-212)."""
        if x.is_cuda and self.config.d_conv == 4:
            return dwconv1d_silu(x, self.conv1d.weight, self.conv1d.bias)      # one fused kernel each way
        L, k = x.shape[1], self.config.d_conv
        w = self.conv1d.weight                                   # (ED, 1, k)
        xp = F.pad(x, (0, 0, k - 1, 0))
        y = xp[:, 0:L] * w[:, 0, 0]
        for j in range(1, k):
            y = y + xp[:, j:j + L] * w[:, 0, j]
        if self.conv1d.bias is not None:
            y = y + self.conv1d.bias
        return F.silu(y)

    def forward(self, x):
        # x : (B, L, D) -> (B, L, D)                               mamba.py:197-225
        if _fused.usable(self, x) and os.environ.get("GFE_MAMBA_UNFUSED") != "1":
            return _fused.mamba_block(self, x)                   # the whole block as one autograd node (gfe_hip/mamba_block.py)
        xz = self.in_proj(x)
        xs, z = xz.chunk(2, dim=-1)
        xs = self._conv_silu(xs)
        y = self.ssm(xs, z)
        return self.out_proj(y)

    def ssm(self, x, z):
        # mamba.py:227-263, plug-in branch: the scan kernel fuses softplus(delta + bias), D*x and y*silu(z)
        cfg = self.config
        A = -torch.exp(self.A_log.float())
        deltaBC = self.x_proj(x)
        delta, B, C = torch.split(deltaBC, [cfg.dt_rank, cfg.d_state, cfg.d_state], dim=-1)
        delta, B, C = self._apply_layernorms(delta, B, C)       # mamba.py:237 (identity unless inner_layernorms)
        delta = _dt_linear(delta, self.dt_proj)
        return selective_scan_tm(x, delta, A, B, C, self.D.float(), z=z, delta_bias=self.dt_proj.bias.float(), delta_softplus=True)

    def _apply_layernorms(self, dt, B, C):
        """mamba.py:188-195: RMSNorm over the dt_rank / d_state axes of the x_proj outputs (Jamba's inner_layernorms)."""
        if self.dt_layernorm is not None:
            dt = self.dt_layernorm(dt)
        if self.B_layernorm is not None:
            B = self.B_layernorm(B)
        if self.C_layernorm is not None:
            C = self.C_layernorm(C)
        return dt, B, C

    def selective_scan(self, x, delta, A, B, C, D):
        """The reference's parallel-scan form (mamba.py:265-286): y = (pscan(exp(delta (x) A), delta B x) @ C) + D x with delta already
        softplus-ed.  x, delta: (B, L, ED); A: (ED, N); B, C: (B, L, N); D: (ED) -> (B, L, ED).  One fused kernel here."""
        return selective_scan_tm(x, delta, A, B, C, D)

    def selective_scan_seq(self, x, delta, A, B, C, D):
        """The reference's sequential definition (mamba.py:288-318) -- the same function of its inputs, same kernel."""
        return selective_scan_tm(x, delta, A, B, C, D)

    # ---- single-token inference (mamba.py:342-405); not on the classify path, plain torch ops --------------------------
    def step(self, x, cache):
        h, inputs = cache
        xz = F.linear(x, self.in_proj.weight, self.in_proj.bias)
        x, z = xz.chunk(2, dim=1)
        x_cache = x.unsqueeze(2)
        x = F.conv1d(torch.cat([inputs, x_cache], dim=2), self.conv1d.weight, self.conv1d.bias,
                     groups=self.config.d_inner)[:, :, 0]
        x = F.silu(x)
        y, h = self.ssm_step(x, h)
        output = F.linear(y * F.silu(z), self.out_proj.weight, self.out_proj.bias)
        inputs = torch.cat([inputs[:, :, 1:], x_cache], dim=2)
        return output, (h, inputs)

    def ssm_step(self, x, h):
        A = -torch.exp(self.A_log.float())
        D = self.D.float()
        deltaBC = F.linear(x, self.x_proj.weight)
        delta, B, C = torch.split(deltaBC, [self.config.dt_rank, self.config.d_state, self.config.d_state], dim=-1)
        delta, B, C = self._apply_layernorms(delta, B, C)
        delta = F.softplus(F.linear(delta, self.dt_proj.weight, self.dt_proj.bias))
        deltaA = torch.exp(delta.unsqueeze(-1) * A)
        BX = delta.unsqueeze(-1) * B.unsqueeze(1) * x.unsqueeze(-1)
        if h is None:
            h = torch.zeros(x.size(0), self.config.d_inner, self.config.d_state, device=deltaA.device)
        h = deltaA * h + BX
        y = (h @ C.unsqueeze(-1)).squeeze(2) + D * x
        return y, h


def _dt_linear(delta, dt_proj):
    """delta = dt_proj.weight @ delta^T without the bias (mamba.py:238); the bias is added inside the scan kernel."""
    from gfe_hip.train_ops import linear
    return linear(delta, dt_proj.weight, None)


class RMSNorm(nn.Module):
    def __init__(self, d_model: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(d_model))

    def forward(self, x):
        if x.is_cuda:
            return rmsnorm(x, self.weight, self.eps)
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps) * self.weight   # mamba.py:415-416 (step() on CPU)
