"""Minimal `torchvision` for 3_style_translator/training/models.py:4-5,302-351: only
`torchvision.ops.deform_conv2d` is on the hot path (torchvision.models / transforms are used by
training-only code and host glue, outside this path)."""
from . import ops  # noqa: F401
__version__ = "0.15.1+dsu"
