"""LoRACompatibleConv / LoRACompatibleLinear without a LoRA layer attached: plain Conv2d / Linear."""
import torch.nn as nn


class LoRACompatibleConv(nn.Conv2d):
    def __init__(self, *args, lora_layer=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.lora_layer = lora_layer


class LoRACompatibleLinear(nn.Linear):
    def __init__(self, *args, lora_layer=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.lora_layer = lora_layer
