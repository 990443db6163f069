"""`from nerfacc import ...` for instant_nsr/models/neus.py:4 and geometry.py:8."""
from drawingspinup_amd.nsr.render import (ContractionType, OccupancyGrid, accumulate_along_rays,  # noqa: F401
                                          ray_marching, render_weight_from_alpha)
__version__ = "0.3.3+dsu"
