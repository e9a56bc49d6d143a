"""Table encoder of the classification loader -- counterpart of the reference's table/deal_table.py:28-61 (`prepare_table`), same
function name, argument and return dictionary.  Host-side (it runs once per dataset, on strings): numpy instead of the reference's
scikit-learn LabelEncoder / StandardScaler, same results:

  * dropped: the bookkeeping columns of the ADNI/TADPOLE export and every baseline (`*bl*`) column; PTID / EXAMDATE / LABEL go to 'info';
  * a column is categorical when it holds strings and at least one of them contains a letter (`discovery_mix`): missing -> 'NA', then
    label-encoded by the sorted distinct values (LabelEncoder); `num_cat` lists the cardinalities in column order;
  * every other column: to_numeric(errors='coerce'), missing -> 0, then z-scored with the population standard deviation
    (StandardScaler: ddof = 0, a constant column is left centred, not divided).
"""
import re

import numpy as np
import pandas as pd

DROP_LIST = ['RID', 'D2', 'SITE', 'DX', 'COLPROT', 'ORIGPROT', 'Month', 'M', 'FDG', 'PIB', 'AV45']      # deal_table.py:30-31
INFO_LIST = ['PTID', 'EXAMDATE', 'LABEL']                                                             # :32


def has_letters(string):
    """deal_table.py:6-14."""
    return isinstance(string, str) and re.search(r'[a-zA-Z]', string) is not None


def discovery_mix(df):
    """Columns of dtype object in which some value contains a letter (deal_table.py:16-25)."""
    return [c for c in df.select_dtypes(include='object').columns if df[c].apply(has_letters).sum() > 0]


def prepare_table(mri_df):
    drop = list(DROP_LIST) + [c for c in mri_df.columns if 'bl' in c]                                  # :33-35
    table_info = mri_df[INFO_LIST]
    df = mri_df.drop(drop + INFO_LIST, axis=1).copy()
    mixed = discovery_mix(df)
    num_columns = [c for c in df.columns if c not in mixed]
    num_cat = []
    for col in df.columns:
        if col in mixed:
            vals = df[col].where(df[col].notna(), 'NA').to_numpy(dtype=object)                         # fillna('NA') (:40)
            uniq, inv = np.unique(vals, return_inverse=True)                                           # LabelEncoder: sorted distinct values
            df[col] = inv.astype(np.int64)
            num_cat.append(len(uniq))
        else:
            df[col] = pd.to_numeric(df[col], errors='coerce').fillna(0)                                # :47-48
    if num_columns:
        x = df[num_columns].to_numpy(dtype=np.float64)
        mean, std = x.mean(0), x.std(0)                                                                # StandardScaler: ddof = 0
        std = np.where(std < 10 * np.finfo(np.float64).eps * np.abs(mean).clip(min=1e-300), 1.0, std)  # constant column: scale 1 (sklearn _handle_zeros_in_scale)
        std = np.where(std == 0.0, 1.0, std)
        df[num_columns] = (x - mean) / std
    return {"info": table_info, "cate_x": df[mixed], "conti_x": df.drop(mixed, axis=1), "num_cat": num_cat, "num_cont": len(num_columns)}
