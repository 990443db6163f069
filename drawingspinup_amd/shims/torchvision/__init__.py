"""`import torchvision` for 3_style_translator/training/models.py:4-5,302-351 and
custom_transforms.py:2.

Only `torchvision.ops.deform_conv2d` is on the hot path.  This package must not take the rest of
torchvision away from the process (transformers / diffusers probe it, and the reference's own
files import `torchvision.models` and `torchvision.transforms`):

  * if a REAL torchvision is installed further down sys.path, it is loaded under the name
    `torchvision` in place of this package and only `torchvision.ops.deform_conv2d` is rebound to
    the gfx950 kernel;
  * otherwise (this image: torchvision is absent and not installable) the package provides
    `ops.deform_conv2d` plus the few names the reference's files use: `models.vgg19` (the
    published layer list; ImageNet weights via a local file, see models.py) and
    `transforms.{Compose, ToTensor, Normalize}` (published semantics, PIL/numpy host code).
"""
import importlib.machinery
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_SHIMS = os.path.dirname(_HERE)


def _find_real():
    paths = [p for p in sys.path if os.path.abspath(p or ".") != _SHIMS]
    try:
        return importlib.machinery.PathFinder.find_spec("torchvision", paths)
    except (ImportError, ValueError):
        return None


_real = _find_real()
if _real is not None and _real.origin and os.path.dirname(os.path.abspath(_real.origin)) != _HERE:
    _mod = importlib.util.module_from_spec(_real)
    sys.modules["torchvision"] = _mod              # `import torchvision` now yields the real one
    _real.loader.exec_module(_mod)
    from drawingspinup_amd.style.generators import deform_conv2d as _dc
    import torchvision.ops as _ops                 # the real subpackage
    _ops.deform_conv2d = _dc
else:
    from . import ops, models, transforms  # noqa: F401
    __version__ = "0.15.1+dsu"
