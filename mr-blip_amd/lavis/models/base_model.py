"""BaseModel: device property, checkpoint loading, from_pretrained/default_config_path — the contract of
lavis/models/base_model.py:19-118 that train.py / evaluate.py / the task and runner rely on."""
import logging
import os

import torch
import torch.nn as nn

from lavis.common.config import load_yaml
from lavis.common.registry import registry


class BaseModel(nn.Module):
    PRETRAINED_MODEL_CONFIG_DICT = {}

    @property
    def device(self):
        return getattr(self, "_device", torch.device("cpu"))

    def load_checkpoint(self, url_or_filename):
        """checkpoint["model"] -> non-strict load (base_model.py:29-56).  URLs are not fetchable here (no network)."""
        if not os.path.isfile(url_or_filename):
            raise RuntimeError("checkpoint url or path is invalid")
        checkpoint = torch.load(url_or_filename, map_location="cpu", weights_only=True)  # tensors / plain containers only
        state_dict = checkpoint["model"] if "model" in checkpoint else checkpoint
        msg = self.load_state_dict(state_dict, strict=False)
        logging.info("Missing keys {}".format(msg.missing_keys))
        logging.info("load checkpoint from %s" % url_or_filename)
        return msg

    @classmethod
    def from_pretrained(cls, model_type):
        model_cfg = load_yaml(cls.default_config_path(model_type)).model
        return cls.from_config(model_cfg)

    @classmethod
    def default_config_path(cls, model_type):
        assert model_type in cls.PRETRAINED_MODEL_CONFIG_DICT, "Unknown model type {}".format(model_type)
        return os.path.join(registry.get_path("library_root"), cls.PRETRAINED_MODEL_CONFIG_DICT[model_type])

    def load_checkpoint_from_config(self, cfg, **kwargs):
        if cfg.get("load_finetuned", False):
            assert cfg.get("finetuned"), "Found load_finetuned is True, but finetune_path is None."
            self.load_checkpoint(cfg.finetuned)
        elif cfg.get("pretrained") and os.path.isfile(str(cfg.pretrained)):
            self.load_from_pretrained(cfg.pretrained)

    def show_n_params(self, return_str=True):
        tot = sum(p.numel() for p in self.parameters())
        return "{:.1f}M".format(tot / 1e6) if return_str else tot
