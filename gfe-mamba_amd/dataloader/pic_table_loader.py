"""The classification dataset of the reference's loader (dataloader/pic_table_loader.py:46-133): MRI volumes named
`PTID-YYYY_MM_DD-L.nii.gz` (README.md:62-69: L = 1 converted / 0 stable) matched against the TADPOLE-derived table by patient id, label and
the nearest examination date, filtered by `days_threshold`, the table encoded by table.deal_table.prepare_table.  Same class / function
names, constructor arguments and sample dictionary (`image` (1, d, h, w) f32, `label`, `cate_x` int64, `conti_x` f32, `name`), so
`classi_dataloader(...)` feeds gfe_hip.step.ClassifyStep the way it feeds the reference's loop (classify_mamba.py:21-24, 94-99).

What differs, deliberately:
  * the image transform runs on the GPU: utils.data_normalization.load_transform = adaptive_normal (bit-exact radix select) -> channel
    first -> Resized(desired_shape) (monai's default mode "area") -> first channel, instead of the monai Compose on the host;
  * volumes are read from `.npy` files (raw (D, H, W) or (D, H, W, C) arrays) as well as `.nii.gz`: NIfTI decoding needs nibabel, which is
    not a dependency of this build -- with nibabel importable `.nii` / `.nii.gz` files load as in the reference (get_fdata), without it they
    raise.  The file-name convention and everything derived from it is the same for both extensions;
  * the file list is SORTED (glob order is file-system dependent, and the reference's filter loop -- kept bug for bug, see __init__ --
    depends on it): with a sorted listing both implementations keep the same files (tests/golden/t11_dataset.json, generated from the
    reference with `glob` returning sorted names).
"""
import re
from datetime import datetime
from glob import glob
from os.path import basename, join

import numpy as np
import pandas as pd
import torch
from torch.utils import data

from table.deal_table import prepare_table

_EXT = (".nii.gz", ".nii", ".npy")


def date_difference(date1, date2):
    """utils/common.py:28-40: |date2 - date1| in days, both 'YYYY-MM-DD'."""
    return abs(datetime.strptime(date2, '%Y-%m-%d') - datetime.strptime(date1, '%Y-%m-%d')).days


def _load_volume(path):
    if path.endswith(".npy"):
        return np.load(path)
    try:
        import nibabel as nib
    except ImportError as e:
        raise RuntimeError(f"{path}: reading NIfTI needs nibabel (not a dependency of this build); convert the volume to .npy "
                           "(np.save of nibabel's get_fdata()) or install nibabel") from e
    return nib.load(path).get_fdata()


def read_nii(ni_path, desired_shape=(160, 160, 96)):
    """pic_table_loader.py:25-43: trilinear (order 1) zoom of a volume to desired_shape on the host (unused by MRI_classify, kept for callers)."""
    from scipy.ndimage import zoom
    vol = _load_volume(ni_path)
    return zoom(vol, tuple(desired_shape[i] / vol.shape[i] for i in range(3)), order=1)


class MRI_classify(data.Dataset):
    def __init__(self, data_path, table_path='', desired_shape=(160, 160, 96), days_threshold=-1, device="cuda"):
        super().__init__()
        self.mri_nii = sorted(p for ext in _EXT[:1] + _EXT[2:] for p in glob(join(data_path, '*' + ext)))
        self.desired_shape, self.device = tuple(desired_shape), device
        self.import_table = len(table_path)
        if self.import_table:
            self.table_df = pd.read_csv(table_path)
            # pic_table_loader.py:67-75, kept as written: the list is popped WHILE it is enumerated (the element after a removed one is
            # skipped), an unmatched file (min_index = -1) is additionally judged by the date_diff of the table's LAST row, and a file
            # that fails both tests pops two entries.  Which files survive is part of the reference's behaviour.
            for i, path in enumerate(self.mri_nii):
                found, min_index = self.find_index(mri_path=basename(path), to_find_table=self.table_df)
                if not found:
                    self.mri_nii.pop(i)
                if self.table_df.iloc[min_index]['date_diff'] <= days_threshold:
                    self.mri_nii.pop(i)
            self.table_df = prepare_table(self.table_df)

    def find_row(self, ID, current_datetime, ischanged, to_find_table):
        """pic_table_loader.py:79-102: among the patient's rows whose LABEL equals the file's label, the one whose EXAMDATE is nearest to
        the scan date, if that is less than 31 days away (ties: the first; a same-day row ends the search)."""
        subset = to_find_table[to_find_table['PTID'] == ID]
        best, min_index = 31, -1
        for index, row in subset.iterrows():
            if not pd.isna(row['LABEL']) and ((ischanged == '1' and int(row['LABEL']) == 1) or (ischanged == '0' and int(row['LABEL']) == 0)):
                dd = date_difference(row['EXAMDATE'], current_datetime)
                if best > dd:
                    best, min_index = dd, index
            if best == 0:
                break
        return (best != 31, min_index)

    def find_index(self, mri_path, to_find_table=None):
        """pic_table_loader.py:119-124: 'PTID-YYYY_MM_DD-L.ext' -> (found, table row)."""
        ID, date, ischanged = mri_path.split('-')
        ischanged = str(ischanged.split('.')[0])
        y, m, d = date.split('_')[:3]
        return self.find_row(ID, f"{y}-{m}-{d}", ischanged, to_find_table)

    def __getitem__(self, index):
        from utils.data_normalization import load_transform
        mri_path = self.mri_nii[index]
        vol = torch.from_numpy(np.ascontiguousarray(_load_volume(mri_path), dtype=np.float32)).to(self.device)
        batch = {'image': load_transform(vol, self.desired_shape)}                                       # :106-110 on the device
        batch['label'] = int(re.findall(r'-(\d)\.(?:nii\.gz|nii|npy)$', mri_path)[0])                     # :111
        if self.import_table:
            _, date_index = self.find_index(basename(mri_path), self.table_df['info'])
            batch['cate_x'] = torch.tensor(self.table_df['cate_x'].iloc[date_index].values, dtype=torch.int64)
            batch['conti_x'] = torch.tensor(self.table_df['conti_x'].iloc[date_index].values, dtype=torch.float32)
        batch['name'] = basename(mri_path)
        return batch

    def __len__(self):
        return len(self.mri_nii)


def classi_dataloader(updir, image_size, batch_size, table_path, shuffle=True, **kwargs):
    """pic_table_loader.py:131-133."""
    dataset = MRI_classify(updir, table_path, image_size, **kwargs)
    return data.DataLoader(dataset, batch_size, shuffle=shuffle, drop_last=True)
