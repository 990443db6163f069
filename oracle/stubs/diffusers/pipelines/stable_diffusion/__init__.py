from dataclasses import dataclass
from typing import Any, Optional

from ...utils import BaseOutput


@dataclass
class StableDiffusionPipelineOutput(BaseOutput):
    images: Any
    nsfw_content_detected: Optional[Any]
