"""MI355X build of the loader's intensity normalisation (reference: utils/data_normalization.py:20-48).  Same function name and
contract -- `adaptive_normal(img)` returns the volume scaled to [-1, 1] by its 0.1 % / 99.9 % quantiles of the non-negative voxels --
but on the device: a 3-pass radix select (gfe_adaptive_normal) instead of a full sort, bit-exact with the reference on f32 input.
Additive: leading batch dimensions are normalised independently when `batched=True` (the reference is called once per volume)."""
import torch

from gfe_hip import call, lib, ptr, stream


def adaptive_normal(img, batched=False, check=True):
    """img: f32 CUDA tensor, any shape (one volume), or (B, ...) with batched=True.  Returns a new tensor (the reference also returns
    a new one: `imgArray = (imgArray - mean) / stddev`).  check=True raises, like the reference's IndexError, when a volume has no
    voxel >= 0 -- the only host synchronisation; pass check=False inside a pipelined loader."""
    if not (img.is_cuda and img.dtype == torch.float32):
        raise TypeError("adaptive_normal: f32 CUDA tensor expected (there is no CPU path)")
    x = img.contiguous()
    B = x.shape[0] if batched else 1
    n = x.numel() // B
    words = lib().gfe_adaptive_normal_ws_words()
    ws = torch.empty(B * words, dtype=torch.int32, device=x.device)
    y = torch.empty_like(x)
    call("gfe_adaptive_normal", ptr(x), ptr(y), ptr(ws), B, n, stream())
    if check and bool((ws.view(B, words)[:, 0] == 0).any()):
        raise IndexError("adaptive_normal: a volume has no voxel >= 0 (index -1 is out of bounds for dimension 0 with size 0)")
    y.an_stats = ws.view(B, words)[:, :8]          # m, ranks, prefixes, lo / hi bit patterns: for tests and logging
    return y


def resize_area(img, size):
    """The loader's `Resized(keys=['image'], spatial_size=desired_shape)` (dataloader/pic_table_loader.py:58) on the device: monai's
    default interpolation mode is "area", i.e. torch.nn.functional.interpolate(mode="area") = adaptive average pooling.
    img: (..., D, H, W) f32 CUDA tensor (leading dimensions are independent volumes / channels) -> (..., d, h, w)."""
    if not (img.is_cuda and img.dtype == torch.float32 and img.dim() >= 3):
        raise TypeError("resize_area: f32 CUDA tensor (..., D, H, W) expected (there is no CPU path)")
    x = img.contiguous()
    D, H, W = x.shape[-3:]
    d, h, w = (int(v) for v in size)
    B = x.numel() // (D * H * W)
    y = torch.empty(x.shape[:-3] + (d, h, w), dtype=torch.float32, device=x.device)
    call("gfe_resize_area", ptr(x), ptr(y), B, D, H, W, d, h, w, stream())
    return y


def load_transform(img, desired_shape=(160, 160, 96)):
    """What MRI_classify.__getitem__ does to a loaded volume (pic_table_loader.py:104-110) on the device: adaptive_normal, channel
    first, Resized(desired_shape), first channel.  img: (D, H, W) or (D, H, W, C) f32 CUDA tensor -> (1, d, h, w)."""
    x = adaptive_normal(img)
    x = x.unsqueeze(0) if x.dim() == 3 else x.movedim(-1, 0)          # EnsureChannelFirstd
    return resize_area(x, desired_shape)[:1]
