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


def test_mri_classify_samples_and_loader_from_npy_volumes(tmp_path):
    """MRI_classify.__getitem__ / classi_dataloader (dataloader/pic_table_loader.py:104-117, 131-133) on .npy volumes: the sample dictionary
    of the reference (image (1, d, h, w) f32 normalised + resized on the device, label from the file name, cate_x / conti_x of the matched
    table row, name) and batches a ClassifyStep can take."""
    import json
    import os
    import numpy as np
    from conftest import GOLDEN
    from dataloader.pic_table_loader import MRI_classify, classi_dataloader
    from utils.data_normalization import load_transform
    fx = json.load(open(os.path.join(GOLDEN, "t11_dataset.json")))
    g = np.random.default_rng(3)
    names = [n.replace(".nii.gz", ".npy") for n in fx["names"][:12]]
    vols = {}
    for n in names:
        v = (np.abs(g.standard_normal((20, 24, 12))) ** 3 * 300).astype(np.float32)
        v[g.random(v.shape) < 0.3] = 0.0
        np.save(tmp_path / n, v)
        vols[n] = v
    csv = os.path.join(GOLDEN, "t11_table_input.csv")
    ds = MRI_classify(str(tmp_path), csv, (16, 16, 8), days_threshold=-1)
    assert 0 < len(ds) <= len(names)
    s = ds[0]
    assert set(s) == {"image", "label", "cate_x", "conti_x", "name"}
    assert s["image"].shape == (1, 16, 16, 8) and s["image"].dtype == torch.float32 and s["image"].is_cuda
    want = load_transform(torch.from_numpy(vols[s["name"]]).cuda(), (16, 16, 8))
    assert torch.equal(s["image"], want)
    assert s["label"] == int(s["name"].split("-")[-1][0]) and s["cate_x"].dtype == torch.int64 and s["conti_x"].dtype == torch.float32
    found, idx = ds.find_index(s["name"], ds.table_df["info"])
    assert torch.equal(s["cate_x"], torch.tensor(ds.table_df["cate_x"].iloc[idx].values, dtype=torch.int64))
    loader = classi_dataloader(str(tmp_path), (16, 16, 8), 2, csv, shuffle=False)
    batch = next(iter(loader))
    assert batch["image"].shape == (2, 1, 16, 16, 8) and batch["cate_x"].shape[0] == 2 and batch["conti_x"].shape[0] == 2 and len(batch["name"]) == 2
    assert batch["image"].abs().max().item() <= 1.0                       # adaptive_normal clips to [-1, 1] (data_normalization.py:44-46)
