"""xformers stand-in (TEST INFRASTRUCTURE ONLY; see oracle/stubs/README.md)."""
from . import ops  # noqa: F401

__version__ = "0.0.17+stub"
