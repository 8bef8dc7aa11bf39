"""MI355X-native (gfx950) ViT / CaiT training hot path.

Host side mirrors the reference's operator interface (create_model / init / apply,
/root/reference/models/create_model.py:6-8, train.py:29-37,82) above a C-ABI HIP library
(include/savit.h, csrc/).  The directory name contains hyphens, so import it with
`importlib.import_module("self-attention-experiments-vision_amd")` or via the alias module
`savit_amd` at the repo root."""
from . import lib  # noqa: F401

__all__ = ["lib"]
