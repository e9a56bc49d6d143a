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
