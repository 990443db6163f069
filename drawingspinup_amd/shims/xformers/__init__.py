"""`import xformers, xformers.ops` for mvdiffusion/models/transformer_mv2d.py:33-36."""
from . import ops  # noqa: F401
__version__ = "0.0.17+dsu"
