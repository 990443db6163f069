"""ModelMixin subset: dtype/device properties and the recursive xformers switch
(`unet.enable_xformers_memory_efficient_attention()`, reference mv.py:186-188)."""
import torch


class ModelMixin(torch.nn.Module):
    config_name = "config.json"
    _supports_gradient_checkpointing = False

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def set_use_memory_efficient_attention_xformers(self, valid, attention_op=None):
        def recurse(module):
            if hasattr(module, "set_use_memory_efficient_attention_xformers"):
                module.set_use_memory_efficient_attention_xformers(valid, attention_op)
            for child in module.children():
                recurse(child)

        for module in self.children():
            if isinstance(module, torch.nn.Module):
                recurse(module)

    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        self.set_use_memory_efficient_attention_xformers(True, attention_op)

    def disable_xformers_memory_efficient_attention(self):
        self.set_use_memory_efficient_attention_xformers(False)


def load_state_dict(*a, **k):
    raise RuntimeError("diffusers stub: no checkpoint access")


def _load_state_dict_into_model(model, state_dict):
    model.load_state_dict(state_dict)
    return []
