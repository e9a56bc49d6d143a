"""Table encoder: table.deal_table.prepare_table against the REFERENCE's own outputs (tests/golden/t5_table.npz, made by
tools/make_golden.py t5 from table/deal_table.py:28-61 on the synthetic TADPOLE-like CSV that travels with it)."""
import os

import numpy as np
import pandas as pd

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
