"""GPU parity of the input normalisation: utils.data_normalization.adaptive_normal (3-pass radix select on the device)
against the reference's own outputs and against the sort-based oracle.  Bit-exact: an order statistic is a value of the input and the
affine map uses the reference's f32 operations in the reference's order."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import ref_ops as O
from test_oracle_golden import AN_CASES

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_adaptive_normal_matches_reference_fixture_bit_for_bit():
    from utils.data_normalization import adaptive_normal
    fx = golden("t4_adaptive_normal.npz")
    for name in AN_CASES:
        y = adaptive_normal(torch.from_numpy(fx[name + ".x"].copy()).cuda()).cpu().numpy()
        want = fx[name + ".y"]
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(y), nan), name
        assert np.array_equal(_bits(y)[~nan], _bits(want)[~nan]), name


@pytest.mark.parametrize("shape,seed", [((96, 96, 96), 0), ((37, 53, 29), 1), ((160, 160, 96), 2), ((5,), 3), ((4099,), 4)])
def test_adaptive_normal_matches_oracle_on_larger_volumes(shape, seed):
    """Odd lengths (scalar tail, unaligned batch rows), the native 160x160x96 volume, skewed intensities with a large zero background
    (the usual MRI histogram: the 0.1 % quantile falls inside the ties at zero)."""
    from utils.data_normalization import adaptive_normal
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g).abs() ** 3 * 400 - 20
    x[torch.rand(shape, generator=g) < 0.35] = 0.0
    y = adaptive_normal(x.cuda())
    want = O.adaptive_normal(x.clone())
    assert torch.equal(y.cpu(), want)
    m = int((x >= 0).sum())
    assert int(y.an_stats[0, 0]) == m


def test_adaptive_normal_batched_and_in_place_semantics():
    from utils.data_normalization import adaptive_normal
    g = torch.Generator().manual_seed(9)
    xs = torch.randn(3, 11, 13, 7, generator=g) * 50 + 10            # 1001 voxels per volume: rows are not 16-B aligned
    keep = xs.clone()
    y = adaptive_normal(xs.cuda(), batched=True)
    assert torch.equal(xs, keep)                                      # the input is not modified
    for b in range(3):
        assert torch.equal(y[b].cpu(), O.adaptive_normal(xs[b].clone())), b


def test_adaptive_normal_empty_selection_raises_like_the_reference():
    from utils.data_normalization import adaptive_normal
    with pytest.raises(IndexError):
        adaptive_normal(torch.full((4, 4, 4), -1.0).cuda())
    with pytest.raises(TypeError):
        adaptive_normal(torch.zeros(4, 4, 4))                         # CPU tensor: no fallback


@pytest.mark.parametrize("shape,size", [((2, 40, 48, 36), (32, 32, 24)), ((1, 256, 256, 166), (160, 160, 96)), ((3, 7, 5, 9), (7, 5, 9)),
                                         ((1, 10, 10, 10), (13, 4, 20)), ((2, 1, 1, 1), (3, 2, 1))])
def test_resize_area_matches_torch_area_interpolation(shape, size):
    """The loader's Resized(spatial_size) (pic_table_loader.py:58; monai's default mode "area" IS F.interpolate(mode="area")): down-,
    up- and mixed resizing, identity, native size 256x256x166 -> 160x160x96.  monai is not in the image: parity is pinned to the torch
    operator monai's Resize calls, on the host."""
    from utils.data_normalization import resize_area
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g) * 100 + 30
    ref = torch.nn.functional.interpolate(x.unsqueeze(0).double(), size=size, mode="area")[0]
    got = resize_area(x.cuda(), size)
    assert tuple(got.shape) == shape[:1] + tuple(size)
    assert (got.double().cpu() - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()
    if tuple(shape[1:]) == tuple(size):
        assert torch.equal(got.cpu(), x)                        # boxes of one voxel: a copy


def test_load_transform_is_normalise_then_resize():
    from utils.data_normalization import adaptive_normal, load_transform, resize_area
    g = torch.Generator().manual_seed(9)
    v = (torch.randn(64, 72, 40, generator=g).abs() * 300).cuda()
    out = load_transform(v, (32, 32, 24))
    assert tuple(out.shape) == (1, 32, 32, 24)
    assert torch.equal(out, resize_area(adaptive_normal(v).unsqueeze(0), (32, 32, 24)))
    assert out.min() >= -1 and out.max() <= 1
