"""diffusers.utils subset."""
import logging as _pylog
from collections import OrderedDict
from dataclasses import fields, is_dataclass

import torch

from . import import_utils  # noqa: F401
from .import_utils import is_xformers_available  # noqa: F401

DIFFUSERS_CACHE = "/tmp/diffusers_stub_cache"
HF_HUB_OFFLINE = True
SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
WEIGHTS_NAME = "diffusion_pytorch_model.bin"


class BaseOutput(OrderedDict):
    """dataclass outputs addressable by attribute, key or index."""

    def __post_init__(self):
        assert is_dataclass(self)
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


def deprecate(*args, **kwargs):
    return None


def maybe_allow_in_graph(cls):
    return cls


def is_torch_version(op, version):
    from packaging.version import parse
    cur = parse(parse(torch.__version__).base_version)
    want = parse(version)
    return {">=": cur >= want, ">": cur > want, "<": cur < want, "<=": cur <= want,
            "==": cur == want}[op]


def is_accelerate_available():
    return False


def is_safetensors_available():
    return True


def _add_variant(weights_name, variant=None):
    return weights_name


def _get_model_file(*a, **k):
    raise RuntimeError("diffusers stub: no hub / checkpoint access")


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    return torch.randn(shape, generator=generator, dtype=dtype).to(device)


class _Logging:
    @staticmethod
    def get_logger(name):
        lg = _pylog.getLogger(name)
        if not hasattr(lg, "warn"):
            lg.warn = lg.warning
        return lg


logging = _Logging()
