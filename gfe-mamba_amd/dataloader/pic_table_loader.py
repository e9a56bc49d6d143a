"""The classification dataset of the reference's loader (dataloader/pic_table_loader.py:46-133): MRI volumes named
`PTID-YYYY_MM_DD-L.nii.gz` (README.md:62-69: L = 1 converted / 0 stable) matched against the TADPOLE-derived table by patient id, label and
the nearest examination date, filtered by `days_threshold`, the table encoded by table.deal_table.prepare_table.  Same class / function
names, constructor arguments and sample dictionary (`image` (1, d, h, w) f32, `label`, `cate_x` int64, `conti_x` f32, `name`), so
`classi_dataloader(...)` feeds gfe_hip.step.ClassifyStep the way it feeds the reference's loop (classify_mamba.py:21-24, 94-99).

What differs, deliberately:
  * the image transform runs on the GPU: utils.data_normalization.load_transform = adaptive_normal (bit-exact radix select) -> channel
    first -> Resized(desired_shape) (monai's default mode "area") -> first channel, instead of the monai Compose on the host;
  * volumes are read from `.nii.gz` / `.nii` by a dependency-free NIfTI-1 reader (read_nifti1 below: the values nibabel's get_fdata() returns,
    which is what monai's LoadImaged hands the reference; nibabel itself is not in this image) and from `.npy` files (raw (D, H, W) or
    (D, H, W, C) arrays).  The file-name convention and everything derived from it is the same for every extension;
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


_NIFTI_DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}


def read_nifti1(path):
    """A NIfTI-1 single-file volume (.nii / .nii.gz) as float64, the values nibabel's `get_fdata()` returns (the reference loads its MRI
    through monai's LoadImaged = nibabel, pic_table_loader.py:104-117): voxel order of the file (x fastest, no reorientation), stored value *
    scl_slope + scl_inter unless the slope is 0 / not finite.  Dependency-free: the 348-byte header by hand (either byte order), gzip from
    the standard library.  NIfTI-2, Analyze pairs (.hdr/.img), complex and RGB voxels are refused."""
    import gzip
    import struct
    with (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")) as fh:
        raw = fh.read()
    if len(raw) < 352:
        raise ValueError(f"{path}: shorter than a NIfTI-1 header")
    for e in ("<", ">"):
        if struct.unpack(e + "i", raw[0:4])[0] == 348:
            break
    else:
        raise ValueError(f"{path}: sizeof_hdr is not 348 (NIfTI-2 or not NIfTI)")
    if raw[344:348] not in (b"n+1\0",):
        raise ValueError(f"{path}: magic {raw[344:348]!r}: only single-file NIfTI-1 ('n+1') is read")
    dim = struct.unpack(e + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(e + "2h", raw[70:74])
    vox_offset, slope, inter = struct.unpack(e + "3f", raw[108:120])
    if datatype not in _NIFTI_DTYPES or not 1 <= dim[0] <= 7:
        raise ValueError(f"{path}: unsupported NIfTI datatype {datatype} / dim[0] = {dim[0]}")
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    dt = np.dtype(e + _NIFTI_DTYPES[datatype])
    if dt.itemsize * 8 != bitpix:
        raise ValueError(f"{path}: bitpix {bitpix} does not match datatype {datatype}")
    n = int(np.prod(shape))
    off = int(vox_offset) if vox_offset >= 352 else 352
    if len(raw) < off + n * dt.itemsize:
        raise ValueError(f"{path}: {len(raw)} bytes, the header promises {off + n * dt.itemsize}")
    vol = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order="F").astype(np.float64)
    if np.isfinite(slope) and slope != 0.0 and np.isfinite(inter) and not (slope == 1.0 and inter == 0.0):
        vol = vol * float(slope) + float(inter)
    return vol


def _load_volume(path):
    if path.endswith(".npy"):
        return np.load(path)
    return read_nifti1(path)


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
