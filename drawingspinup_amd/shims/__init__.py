"""Drop-in packages for the five import names through which the reference reaches native code
(SURVEY.md §8b).  `install()` puts this directory first on sys.path so that the reference's
unmodified `import tinycudann as tcnn`, `from nerfacc import ...`, `import xformers.ops`,
`import torchvision` resolve to the gfx950 kernels."""
import os
import sys


def install():
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    return here
