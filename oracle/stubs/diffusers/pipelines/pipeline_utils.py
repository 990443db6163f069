"""DiffusionPipeline subset the reference's pipeline relies on: register_modules,
_execution_device, progress_bar, set_progress_bar_config (diffusers 0.19.3 pipeline_utils.py)."""
import contextlib

import torch

from ..configuration_utils import ConfigMixin


class _Bar:
    def update(self, n=1):
        pass


class DiffusionPipeline(ConfigMixin):
    config_name = "model_index.json"

    def register_modules(self, **kwargs):
        for name, module in kwargs.items():
            self.register_to_config(**{name: (None, None) if module is None else ("stub", type(module).__name__)})
            setattr(self, name, module)

    @property
    def device(self):
        for v in vars(self).values():
            if isinstance(v, torch.nn.Module):
                return next(v.parameters()).device
        return torch.device("cpu")

    @property
    def _execution_device(self):
        return self.device

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    @contextlib.contextmanager
    def progress_bar(self, iterable=None, total=None):
        yield _Bar()
