"""Autograd node of Jamba's SparseMoEBlock (cross_atten/jamba.py:441-535) over the grouped kernels of csrc/moe.hip: device-side routing +
expert sort, one grouped exact-f32 GEMM per projection for ALL experts, gather combine.  ~8 launches forward, ~14 backward per MoE layer
whatever the number of experts, no host synchronisation (the reference walks the experts in Python and reads the hit counts on the host)."""
import torch

from . import call, ptr, stream
from . import nn_ops as K
from .train_ops import _grad_slot

F32 = torch.float32
_TABLES = {}


def _ptr_table(tensors, dev):
    """device array of the tensors' addresses, cached on the addresses themselves (parameters owned by FlatAdam never move)"""
    key = tuple(t.data_ptr() for t in tensors)
    tab = _TABLES.get(key)
    if tab is None:
        tab = torch.tensor(key, dtype=torch.int64).to(dev)
        if len(_TABLES) > 256:
            _TABLES.clear()
        _TABLES[key] = tab
    return tab


class _MoEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, top_k, router_w, *expert_w):
        E = len(expert_w) // 3
        gate, up, down = expert_w[:E], expert_w[E:2 * E], expert_w[2 * E:]
        T, D = x.shape
        Fd = gate[0].shape[0]
        P = T * top_k
        dev = x.device
        x2 = x.detach()
        if x2.dtype != F32 or not x2.is_contiguous():
            x2 = x2.float().contiguous()
        det = lambda t: t.detach()
        logits = K.gemm_f32(x2, False, det(router_w), False)                                # (T, E)                       jamba.py:484
        rw = torch.empty((T, top_k), dtype=F32, device=dev)
        ints = torch.empty(T * top_k * 3 + E + 1, dtype=torch.int32, device=dev)
        sel, tok_sorted, pos, seg = torch.split(ints, [P, P, P, E + 1])
        call("gfe_moe_route", ptr(logits), T, E, top_k, ptr(rw), ptr(sel), ptr(tok_sorted), ptr(pos), ptr(seg), stream())   # :485-493
        tg, tu, td = (_ptr_table([det(w) for w in ws], dev) for ws in (gate, up, down))
        g = torch.empty((P, Fd), dtype=F32, device=dev)
        u = torch.empty((P, Fd), dtype=F32, device=dev)
        call("gfe_moe_gemm_rows", ptr(x2), D, ptr(tok_sorted), ptr(tg), D, 0, ptr(g), Fd, ptr(seg), E, P, Fd, D, 0, stream())
        call("gfe_moe_gemm_rows", ptr(x2), D, ptr(tok_sorted), ptr(tu), D, 0, ptr(u), Fd, ptr(seg), E, P, Fd, D, 0, stream())
        h = torch.empty_like(g)
        call("gfe_moe_act_fwd", ptr(g), ptr(u), ptr(h), P * Fd, stream())                  # silu(gate) * up                :535
        o = torch.empty((P, D), dtype=F32, device=dev)
        call("gfe_moe_gemm_rows", ptr(h), Fd, None, ptr(td), Fd, 0, ptr(o), D, ptr(seg), E, P, D, Fd, 0, stream())
        out = torch.empty((T, D), dtype=F32, device=dev)
        call("gfe_moe_combine", ptr(o), ptr(rw), ptr(pos), ptr(out), T, top_k, D, 0, stream())   # weighted sum per token      :508-513
        ctx.save_for_backward(x2, logits, rw, sel, tok_sorted, pos, seg, g, u, h, o, router_w, *expert_w)
        ctx.meta = (T, D, Fd, E, top_k)
        return out, logits                                   # the router logits stay differentiable, as in the reference (jamba.py:517)

    @staticmethod
    def backward(ctx, dout, dlogits):
        x2, logits, rw, sel, tok_sorted, pos, seg, g, u, h, o, router_w, *expert_w = ctx.saved_tensors
        T, D, Fd, E, top_k = ctx.meta
        P = T * top_k
        dev = x2.device
        gate, up, down = expert_w[:E], expert_w[E:2 * E], expert_w[2 * E:]
        det = lambda t: t.detach()
        if dout is None:                                     # only the router logits carry a gradient (the balance term alone)
            dout = torch.zeros((T, D), dtype=F32, device=dev)
        d = dout if (dout.dtype == F32 and dout.is_contiguous()) else dout.float().contiguous()
        tg, tu, td = (_ptr_table([det(w) for w in ws], dev) for ws in (gate, up, down))

        def grad_targets(ws):
            """per-expert accumulation targets: the optimizer's own slots when it owns the parameters, else one zeroed buffer"""
            slots = [_grad_slot(w) for w in ws]
            if all(s is not None for s in slots):
                return slots, None, 1
            buf = torch.zeros((len(ws),) + tuple(ws[0].shape), dtype=F32, device=dev)
            return list(buf.unbind(0)), buf, 0

        do = torch.empty((P, D), dtype=F32, device=dev)
        drw = torch.empty((T, top_k), dtype=F32, device=dev)
        call("gfe_moe_combine_bwd", ptr(d), ptr(o), ptr(rw), ptr(pos), ptr(do), ptr(drw), T, top_k, D, stream())
        # down projection
        dh = torch.empty((P, Fd), dtype=F32, device=dev)
        call("gfe_moe_gemm_rows", ptr(do), D, None, ptr(td), Fd, 1, ptr(dh), Fd, ptr(seg), E, P, Fd, D, 0, stream())
        gd, bd, acc_d = grad_targets(down)
        call("gfe_moe_gemm_wgrad", ptr(do), D, ptr(h), Fd, None, ptr(_ptr_table(gd, dev)), Fd, ptr(seg), E, D, Fd, acc_d, stream())
        # activation
        dg, du = torch.empty_like(g), torch.empty_like(u)
        call("gfe_moe_act_bwd", ptr(g), ptr(u), ptr(dh), ptr(dg), ptr(du), P * Fd, stream())
        # gate / up projections
        dxs = torch.empty((P, D), dtype=F32, device=dev)
        call("gfe_moe_gemm_rows", ptr(dg), Fd, None, ptr(tg), D, 1, ptr(dxs), D, ptr(seg), E, P, D, Fd, 0, stream())
        call("gfe_moe_gemm_rows", ptr(du), Fd, None, ptr(tu), D, 1, ptr(dxs), D, ptr(seg), E, P, D, Fd, 1, stream())
        gg, bg, acc_g = grad_targets(gate)
        gu, bu, acc_u = grad_targets(up)
        call("gfe_moe_gemm_wgrad", ptr(dg), Fd, ptr(x2), D, ptr(tok_sorted), ptr(_ptr_table(gg, dev)), D, ptr(seg), E, Fd, D, acc_g, stream())
        call("gfe_moe_gemm_wgrad", ptr(du), Fd, ptr(x2), D, ptr(tok_sorted), ptr(_ptr_table(gu, dev)), D, ptr(seg), E, Fd, D, acc_u, stream())
        dx = torch.empty((T, D), dtype=F32, device=dev)
        call("gfe_moe_combine", ptr(dxs), None, ptr(pos), ptr(dx), T, top_k, D, 0, stream())
        # router: d logits -> d W_r, and its share of dx
        dlog = torch.empty((T, E), dtype=F32, device=dev)
        call("gfe_moe_route_bwd", ptr(logits), ptr(sel), ptr(drw), ptr(dlog), T, E, top_k, stream())
        if dlogits is not None:                              # a loss on the returned router logits (load_balancing_loss, jamba.py:537-556)
            dlog += dlogits.to(F32)
        K.gemm_f32(dlog, False, det(router_w), True, accum_into=dx)
        slot = _grad_slot(router_w)
        if slot is not None:
            K.gemm_f32(dlog, True, x2, True, accum_into=slot)
            d_router = None
        else:
            d_router = K.gemm_f32(dlog, True, x2, True)
        outs = []
        for buf, ws in ((bg, gate), (bu, up), (bd, down)):
            outs += [None] * len(ws) if buf is None else list(buf.unbind(0))
        return (dx, None, d_router, *outs)


def moe_mlp(x2, top_k, router_w, gate_ws, up_ws, down_ws):
    """x2: (T, D) f32 tokens -> (out (T, D), router_logits (T, E)).  Experts: lists of the gate / up (F, D) and down (D, F) weights."""
    if not x2.is_cuda:
        raise RuntimeError("gfe_hip moe_mlp needs CUDA/HIP tensors (no CPU fallback)")
    return _MoEFn.apply(x2, int(top_k), router_w, *gate_ws, *up_ws, *down_ws)
