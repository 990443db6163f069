class UNet2DConditionLoadersMixin:
    """LoRA / attn-processor loading: not on the inference path."""
