"""Model zoo of this build: only the Mr. BLIP family member named by the north star (``blip2_mr``).
``load_model`` / ``registry.get_model_class`` mirror lavis/models/__init__.py:83-120."""
import torch

from lavis.common.registry import registry
from lavis.models.base_model import BaseModel
from lavis.models.blip2_mr_models.blip2_mr import BLIP2_MR

__all__ = ["load_model", "BaseModel", "BLIP2_MR"]


def load_model(name, model_type, is_eval=False, device="cpu", checkpoint=None):
    model = registry.get_model_class(name).from_pretrained(model_type=model_type)
    if checkpoint is not None:
        model.load_checkpoint(checkpoint)
    if is_eval:
        model.eval()
    return model
