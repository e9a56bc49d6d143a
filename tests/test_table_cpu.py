"""Table encoder: table.deal_table.prepare_table against the REFERENCE's own outputs (tests/golden/t5_table.npz, made by
tools/make_golden.py t5 from table/deal_table.py:28-61 on the synthetic TADPOLE-like CSV that travels with it)."""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, golden


def test_prepare_table_matches_the_reference():
    from table.deal_table import discovery_mix, has_letters, prepare_table
    fx = golden("t5_table.npz")
    df = pd.read_csv(os.path.join(GOLDEN, "t5_table_input.csv"))
    out = prepare_table(df)
    assert list(out["cate_x"].columns) == list(fx["cate_cols"]) and list(out["conti_x"].columns) == list(fx["conti_cols"])
    assert list(out["info"].columns) == list(fx["info_cols"]) == ["PTID", "EXAMDATE", "LABEL"]
    assert np.array_equal(out["cate_x"].to_numpy(dtype=np.int64), fx["cate_x"])                    # label codes: exact
    assert out["num_cat"] == fx["num_cat"].tolist() and out["num_cont"] == int(fx["num_cont"])
    assert np.abs(out["conti_x"].to_numpy(dtype=np.float64) - fx["conti_x"]).max() < 1e-12          # z-scores (ddof 0, missing -> 0 first)
    assert "ICV" in out["conti_x"].columns and np.all(out["conti_x"]["ICV"].to_numpy() == 0)         # a constant column is centred, not divided
    assert not any("bl" in c for c in list(out["cate_x"].columns) + list(out["conti_x"].columns))   # baseline columns are dropped
    # '>1700' / '<80' hold no letter: the column stays numeric, the unparsable entries are coerced to NaN and then to 0 (deal_table.py:47-48)
    assert "ABETA" in out["conti_x"].columns and "TAU" in out["conti_x"].columns
    assert has_letters("Male") and not has_letters(">1700") and not has_letters(3.0)
    assert len(df) == len(out["cate_x"]) and "PTGENDER" in discovery_mix(df)


def test_mri_classify_dataset_matches_the_reference_fixture(tmp_path):
    """dataloader/pic_table_loader.py:46-127 (VERDICT r03 missing #3): which files the constructor keeps (the reference's pop-while-enumerating
    filter, days_threshold -1 / 30 / 200), the table row every kept file is matched to, and label / cate_x / conti_x of every sample --
    against tests/golden/t11_dataset.json, produced by the reference's own class on the same file names and CSV (tools/make_golden.py t11)."""
    import json
    import re
    from dataloader.pic_table_loader import MRI_classify
    fx = json.load(open(os.path.join(GOLDEN, "t11_dataset.json")))
    for n in fx["names"]:
        open(tmp_path / n, "wb").close()
    for thr, case in fx["cases"].items():
        ds = MRI_classify(str(tmp_path), os.path.join(GOLDEN, "t11_table_input.csv"), (8, 8, 8), days_threshold=int(thr), device="cpu")
        kept = [os.path.basename(p) for p in ds.mri_nii]
        assert kept == case["kept"], thr
        assert len(ds) == len(case["kept"])
        for n, row, lab, cate, conti in zip(kept, case["rows"], case["labels"], case["cate_x"], case["conti_x"]):
            found, idx = ds.find_index(n, ds.table_df["info"])
            assert [bool(found), int(idx)] == row, n
            assert int(re.findall(r"-(\d)\.nii\.gz$", n)[0]) == lab
            assert [int(v) for v in ds.table_df["cate_x"].iloc[idx].values] == cate
            assert np.allclose(np.asarray(ds.table_df["conti_x"].iloc[idx].values, dtype=np.float64), np.asarray(conti), rtol=0, atol=1e-12)


def _write_nifti1(path, arr, endian="<", slope=0.0, inter=0.0, vox_offset=352.0, gz=False):
    """A minimal single-file NIfTI-1 volume written by hand from the format's header table (sizeof_hdr 348, dim @40, datatype @70, bitpix @72,
    vox_offset @108, scl_slope @112, scl_inter @116, magic 'n+1' @344; voxels x-fastest behind a 4-byte extension flag)."""
    import gzip
    import struct
    codes = {"u1": 2, "i2": 4, "i4": 8, "f4": 16, "f8": 64, "i1": 256, "u2": 512}
    kind = arr.dtype.str[1:]
    hdr = bytearray(348)
    hdr[0:4] = struct.pack(endian + "i", 348)
    dim = [arr.ndim] + list(arr.shape) + [1] * (7 - arr.ndim)
    hdr[40:56] = struct.pack(endian + "8h", *dim)
    hdr[70:74] = struct.pack(endian + "2h", codes[kind], arr.dtype.itemsize * 8)
    hdr[76:108] = struct.pack(endian + "8f", 1, 1, 1, 1, 1, 1, 1, 1)
    hdr[108:120] = struct.pack(endian + "3f", vox_offset, slope, inter)
    hdr[344:348] = b"n+1\0"
    body = bytes(hdr) + b"\0" * (int(vox_offset) - 348) + arr.astype(arr.dtype.newbyteorder(endian)).tobytes(order="F")
    with (gzip.open(path, "wb") if gz else open(path, "wb")) as fh:
        fh.write(body)


def test_nifti1_reader_without_nibabel(tmp_path):
    """VERDICT r05 missing #5: `.nii` / `.nii.gz` volumes load without nibabel (dataloader/pic_table_loader.py read_nifti1), with the values
    get_fdata() defines: file voxel order (x fastest), float64, stored * scl_slope + scl_inter unless the slope is 0; both byte orders, gzip,
    a non-default vox_offset; unsupported files raise instead of returning garbage."""
    from dataloader.pic_table_loader import read_nifti1, read_nii, _load_volume
    rng = np.random.default_rng(5)
    a16 = rng.integers(-3000, 3000, size=(5, 7, 3)).astype(np.int16)
    af = rng.standard_normal((4, 6, 5)).astype(np.float32)
    au = rng.integers(0, 255, size=(3, 4, 5, 2)).astype(np.uint8)
    _write_nifti1(tmp_path / "a.nii", a16)
    _write_nifti1(tmp_path / "b.nii.gz", a16, endian=">", slope=0.5, inter=-7.0, gz=True)
    _write_nifti1(tmp_path / "c.nii.gz", af, vox_offset=416.0, gz=True)
    _write_nifti1(tmp_path / "d.nii", au, slope=1.0, inter=0.0)
    v = read_nifti1(str(tmp_path / "a.nii"))
    assert v.dtype == np.float64 and v.shape == a16.shape and np.array_equal(v, a16.astype(np.float64))
    assert np.array_equal(read_nifti1(str(tmp_path / "b.nii.gz")), a16.astype(np.float64) * 0.5 - 7.0)
    assert np.array_equal(read_nifti1(str(tmp_path / "c.nii.gz")), af.astype(np.float64))
    assert np.array_equal(_load_volume(str(tmp_path / "d.nii")), au.astype(np.float64))
    z = read_nii(str(tmp_path / "c.nii.gz"), desired_shape=(8, 12, 10))                    # pic_table_loader.py:25-43 on a decoded volume
    assert z.shape == (8, 12, 10)
    raw = bytearray(open(tmp_path / "a.nii", "rb").read())
    raw[344:348] = b"ni1\0"                                                                 # header/image pair: refused
    open(tmp_path / "bad.nii", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        read_nifti1(str(tmp_path / "bad.nii"))
    open(tmp_path / "short.nii", "wb").write(bytes(raw[:400]))
    with pytest.raises(ValueError):
        read_nifti1(str(tmp_path / "short.nii"))
