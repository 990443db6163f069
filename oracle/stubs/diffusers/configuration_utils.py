"""ConfigMixin / register_to_config / FrozenDict: constructor arguments captured as `self.config`
with attribute access (what `self.config.class_embed_type` etc. rely on,
unet_mv2d_condition.py:857-927)."""
import functools
import inspect


class FrozenDict(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for key, value in self.items():
            object.__setattr__(self, key, value)

    def __setitem__(self, k, v):
        raise TypeError("FrozenDict is read-only")


class ConfigMixin:
    config_name = None

    def register_to_config(self, **kwargs):
        kwargs.pop("kwargs", None)
        cur = dict(getattr(self, "_internal_dict", {}))
        cur.update(kwargs)
        self._internal_dict = FrozenDict(cur)

    @property
    def config(self):
        return self._internal_dict


def register_to_config(init):
    @functools.wraps(init)
    def inner_init(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = [p for n, p in sig.parameters.items() if n != "self"]
        cfg = {p.name: p.default for p in params if p.default is not inspect.Parameter.empty}
        for p, a in zip(params, args):
            cfg[p.name] = a
        cfg.update({k: v for k, v in kwargs.items() if not k.startswith("_")})
        init(self, *args, **{k: v for k, v in kwargs.items() if not k.startswith("_")})
        if isinstance(self, ConfigMixin):
            merged = dict(cfg)
            merged.update(dict(getattr(self, "_internal_dict", {})))   # explicit calls inside __init__ win
            self.register_to_config(**merged)
    return inner_init
