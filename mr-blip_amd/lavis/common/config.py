"""Run/model/dataset YAML configuration with ``--options k=v`` overrides — the reference's ``Config`` (lavis/common/config.py:16-126)
over a small attribute-dict instead of OmegaConf (not installed here): same three roots (run, model, datasets), same merge order
(arch default yaml < user model section < user options), same accessors (run_cfg / model_cfg / datasets_cfg / to_dict / pretty_print)."""
import json
import logging
import os
import re

import yaml

from lavis.common.registry import registry


_FLOAT_RE = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$")


def _coerce(v):
    """PyYAML (YAML 1.1) reads ``3e-4`` as a string; OmegaConf reads a float — follow OmegaConf."""
    if isinstance(v, str) and _FLOAT_RE.match(v):
        return float(v)
    if isinstance(v, list):
        return [_coerce(x) for x in v]
    return v


class Node(dict):
    """dict with attribute access and ``get``; nested dicts become Nodes."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, Node(v) if isinstance(v, dict) and not isinstance(v, Node) else _coerce(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Node) else v) for k, v in self.items()}


def merge(*nodes) -> Node:
    out = Node()
    for n in nodes:
        for k, v in (n or {}).items():
            if isinstance(v, dict) and isinstance(out.get(k), dict):
                out[k] = merge(out[k], v)
            else:
                out[k] = Node(v) if isinstance(v, dict) else v
    return out


def load_yaml(path) -> Node:
    with open(path) as f:
        return Node(yaml.safe_load(f) or {})


def from_dotlist(items) -> Node:
    out = Node()
    for it in items or []:
        key, _, val = it.partition("=")
        cur = out
        parts = key.split(".")
        for p in parts[:-1]:
            if p not in cur:
                cur[p] = Node()
            cur = cur[p]
        cur[parts[-1]] = yaml.safe_load(val)
    return out


class Config:
    def __init__(self, args):
        self.args = args
        registry.register("configuration", self)
        opts = self._convert_to_dot_list(getattr(args, "options", None))
        user = from_dotlist(opts)
        config = load_yaml(args.cfg_path)
        runner_config = Node({"run": config.get("run", {})})
        model_config = self.build_model_config(config, user)
        dataset_config = self.build_dataset_config(config)
        self.config = merge(runner_config, model_config, dataset_config, user)

    @staticmethod
    def build_model_config(config, user=None):
        model = config.get("model")
        assert model is not None, "Missing model configuration file."
        model_cls = registry.get_model_class(model.arch)
        assert model_cls is not None, f"Model '{model.arch}' has not been registered."
        model_type = (user or {}).get("model", {}).get("model_type") or model.get("model_type")
        assert model_type is not None, "Missing model_type."
        default = load_yaml(model_cls.default_config_path(model_type=model_type))
        return merge(default, {"model": model})

    @staticmethod
    def build_dataset_config(config):
        datasets = config.get("datasets")
        if datasets is None:
            raise KeyError("Expecting 'datasets' as the root key for dataset configuration.")
        out = Node()
        for name in datasets:
            builder_cls = registry.get_builder_class(name)
            assert builder_cls is not None, f"Dataset builder '{name}' has not been registered."
            default = load_yaml(builder_cls.default_config_path(type=datasets[name].get("type", "default")))
            out = merge(out, default, {"datasets": {name: datasets[name]}})
        return out

    @staticmethod
    def _convert_to_dot_list(opts):
        if not opts:
            return []
        if opts[0].find("=") != -1:
            return list(opts)
        return [k + "=" + v for k, v in zip(opts[0::2], opts[1::2])]

    def get_config(self):
        return self.config

    @property
    def run_cfg(self):
        return self.config.run

    @property
    def datasets_cfg(self):
        return self.config.datasets

    @property
    def model_cfg(self):
        return self.config.model

    def pretty_print(self):
        logging.info("\n=====  Running Parameters    =====")
        logging.info(json.dumps(self.config.run.to_dict(), indent=4, sort_keys=True))
        logging.info("\n======  Dataset Attributes  ======")
        for d in self.config.datasets:
            logging.info(f"\n======== {d} =======")
            logging.info(json.dumps(self.config.datasets[d].to_dict(), indent=4, sort_keys=True))
        logging.info("\n======  Model Attributes  ======")
        logging.info(json.dumps(self.config.model.to_dict(), indent=4, sort_keys=True))

    def to_dict(self):
        return self.config.to_dict()
