"""Drop-in at the reference's default geometry: the three modules built with the constructor calls of classify_mamba.py:36-56 must have
exactly the state-dict keys, shapes and dtypes of the reference's (so that the authors' model.pt / model_current.pth /
ft_model_current.pth load).  The listing in tests/golden/t7_native_state_dict.json was written by tools/make_golden.py t7 from the
imported reference; nothing is allocated here (meta device)."""
import json
import os

import torch

from conftest import GOLDEN


def _listing(m):
    return {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}


def test_default_constructors_give_the_references_state_dict_layout():
    from classify.classifier import Combine_classfier_vit_mid
    from cross_atten.mamba_transformer import Cross_mamba_both
    from pytorch3dunet.unet3d.model import Residual_mid_UNet3D_vit
    ref = json.load(open(os.path.join(GOLDEN, "t7_native_state_dict.json")))
    with torch.device("meta"):
        gen = Residual_mid_UNet3D_vit(1, 1, is_segmentation=False, f_maps=(64, 128, 256))
        head = Combine_classfier_vit_mid(seq_length=4)
        ft = Cross_mamba_both(categories=(11, 2, 2, 4, 4, 3, 3), num_continuous=25, dim=512, dim_out=1, depth=6, heads=8,
                              attn_dropout=0.1, ff_dropout=0.1, dim_head=512 // 8)
    for name, m in (("gen", gen), ("head", head), ("ft", ft)):
        ours = _listing(m)
        assert set(ours) == set(ref[name]), (name, sorted(set(ours) ^ set(ref[name])))
        for k, v in ref[name].items():
            assert ours[k] == v, (name, k, ours[k], v)
    assert len(ref["gen"]) == 104 and len(ref["head"]) == 2 and len(ref["ft"]) == 83
    # the geometry the reference hard-codes (model.py:107-117, classifier.py:327, mamba_transformer.py:84)
    assert ref["gen"]["mid.to_patch_embedding.2.weight"][0] == [512, 40 * 40 * 256]
    assert ref["head"]["vit_mid_linear.weight"][0] == [4, 320 * 120]
    assert ref["ft"]["final_cross.k_proj.weight"][0] == [512, 160 * 160]
