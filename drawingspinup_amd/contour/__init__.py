"""Contour-remover stage: the FFC-ResNet generator of
1_lama_contour_remover, restated with the reference's state_dict layout."""
from .ffc import FFCResNetGenerator, make_generator, LAMA_FOURIER_GENERATOR   # noqa: F401
