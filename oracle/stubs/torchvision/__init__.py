"""torchvision stand-in (TEST INFRASTRUCTURE ONLY): transforms.functional.to_pil_image / to_tensor
as torchvision 0.15.1 defines them (reference README.md:29; call sites
pipeline_mvdiffusion_image.py:169,358)."""
from . import transforms  # noqa: F401
