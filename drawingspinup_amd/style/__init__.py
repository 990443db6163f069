"""Per-frame style translator (stage-1 GeneratorJ_RIC, stage-2 GeneratorJ) on gfx950 kernels."""
from .generators import GeneratorJ, GeneratorJ_RIC, build_model, deform_conv2d, generate_coordinates  # noqa: F401
