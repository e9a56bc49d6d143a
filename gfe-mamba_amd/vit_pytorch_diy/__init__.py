"""MI355X build of the one ViT the reference's hot path instantiates (vit_pytorch_diy/__init__.py:1 -> vit.ViT),
plus the API-compatible vit_3d.ViT twin.  The reference's vendored model zoo (cait, mae, dino, ...) is out of scope."""
from vit_pytorch_diy.vit import ViT  # noqa: F401
from vit_pytorch_diy import vit_3d  # noqa: F401
