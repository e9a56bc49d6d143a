"""Stage outputs of the frozen generator and bit-exact fingerprints of them, for the batch-invariance diagnostics
(tests/test_configs_gpu.py, tools/timing_fuzz.py).

staged_forward() runs the PRODUCT forward (`gen(x, output_vit_mid=True)`: same calls, same fused hand-offs) and collects what every stage
returned through forward hooks; fingerprint() turns a stage tensor into one 64-bit integer per 8 x 8 x 8 voxel tile of every sample (or per
4096-word chunk for tensors that are not channels-last volumes), so that two runs can be compared bit for bit without keeping a second
copy of 906 MB tensors, and a difference names the tiles it sits in."""
import torch

STAGES = ("encoders.0", "encoders.1", "encoders.2", "mid", "decoders.0", "decoders.1")


def staged_forward(gen, x):
    """-> {stage name: output tensor} + 'mid_input', 'mid_output', 'pet' as the caller of the generator sees them."""
    got, hooks = {}, []
    mods = dict(gen.named_modules())
    for name in STAGES:
        if name in mods:
            hooks.append(mods[name].register_forward_hook(lambda m, i, o, name=name: got.__setitem__(name, o[0] if isinstance(o, tuple) else o)))
    try:
        with torch.no_grad():
            mi, mo, pet = gen(x, output_vit_mid=True)
    finally:
        for h in hooks:
            h.remove()
    got["mid_input"], got["mid_output"], got["pet"] = mi, mo, pet
    return got


def _words(t):
    """bit pattern of a tensor as int64 words, original shape kept"""
    if t.dtype == torch.bfloat16 or t.dtype == torch.float16:
        return t.view(torch.int16).to(torch.int64)
    if t.dtype == torch.float32:
        return t.view(torch.int32).to(torch.int64)
    raise TypeError(t.dtype)


def fingerprint(t):
    """(B, cells) int64: one wrapped weighted sum per cell of every sample.  Channels-last volumes (B, D, H, W, C) with D, H, W multiples of
    8: cell = conv tile (td, th, tw) in the conv kernels' own order ((td * nth + th) * ntw + tw); anything else: cell = 4096-word chunk."""
    t = t.contiguous()
    B = t.shape[0]
    out = []
    if t.dim() == 5 and t.shape[-1] >= 8 and all(s % 8 == 0 for s in t.shape[1:4]):
        D, H, W, C = t.shape[1:]
        n = 8 * 8 * 8 * C
        wts = ((torch.arange(n, device=t.device, dtype=torch.int64) * 2654435761) % 2147483647) | 1
        wts = wts.view(1, 8, 1, 8, 1, 8, C)
        for b in range(B):
            v = _words(t[b]).view(D // 8, 8, H // 8, 8, W // 8, 8, C)
            out.append((v * wts).sum(dim=(1, 3, 5, 6)).reshape(-1))
    else:
        for b in range(B):
            v = _words(t[b]).reshape(-1)
            pad = (-v.numel()) % 4096
            if pad:
                v = torch.cat([v, v.new_zeros(pad)])
            wts = ((torch.arange(4096, device=t.device, dtype=torch.int64) * 2654435761) % 2147483647) | 1
            out.append((v.view(-1, 4096) * wts).sum(dim=1))
    return torch.stack(out)


def fingerprints(stages):
    return {k: fingerprint(v).cpu() for k, v in stages.items()}


def first_divergence(fp_a, fp_b, order=STAGES + ("mid_input", "mid_output", "pet")):
    """-> None, or (stage name, [(sample, cell), ...] up to 16) for the first stage in network order whose fingerprints differ."""
    for k in order:
        if k not in fp_a or k not in fp_b:
            continue
        ne = (fp_a[k] != fp_b[k]).nonzero()
        if ne.numel():
            return k, [tuple(int(i) for i in r) for r in ne[:16]], int(ne.shape[0])
    return None


def device_report():
    """what distinguishes one box of the pool from another, as far as an ordinary user can see"""
    p = torch.cuda.get_device_properties(0)
    rep = {"name": p.name, "cus": p.multi_processor_count, "gcn_arch": getattr(p, "gcnArchName", "?"), "total_memory_gb": round(p.total_memory / 2**30, 1),
           "clock_rate_khz": getattr(p, "clock_rate", None), "torch": torch.__version__, "hip": torch.version.hip}
    try:
        import subprocess
        out = subprocess.run(["rocm-smi", "--showclocks", "--showperflevel", "--showpower", "--showuniqueid"], capture_output=True, text=True, timeout=20).stdout
        rep["rocm_smi"] = [ln.strip() for ln in out.splitlines() if "GPU[" in ln][:24]
    except Exception as e:  # noqa: BLE001 -- diagnostics only
        rep["rocm_smi"] = repr(e)
    return rep
