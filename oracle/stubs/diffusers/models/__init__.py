from .modeling_utils import ModelMixin  # noqa: F401


class UNet2DConditionModel:      # type annotation only (pipeline_mvdiffusion_image.py:27,77)
    pass


class AutoencoderKL:             # type annotation only; the golden script supplies the instance
    pass
