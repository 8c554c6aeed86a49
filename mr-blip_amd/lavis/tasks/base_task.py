"""BaseTask: build_model / build_datasets / train_step / evaluation loop (contract of lavis/tasks/base_task.py:23-288)."""
import json
import logging
import os

import torch
import torch.distributed as dist

from lavis.common.dist_utils import get_rank, get_world_size, is_dist_avail_and_initialized, is_main_process
from lavis.common.logger import MetricLogger
from lavis.common.registry import registry


class BaseTask:
    def __init__(self, **kwargs):
        self.inst_id_key = "instance_id"

    @classmethod
    def setup_task(cls, **kwargs):
        return cls()

    def build_model(self, cfg):
        model_config = cfg.model_cfg
        model_cls = registry.get_model_class(model_config.arch)
        return model_cls.from_config(model_config)

    def build_datasets(self, cfg):
        datasets = {}
        datasets_config = cfg.datasets_cfg
        assert len(datasets_config) > 0, "At least one dataset has to be specified."
        for name in datasets_config:
            builder = registry.get_builder_class(name)(datasets_config[name])
            datasets[name] = builder.build_datasets()
        return datasets

    def train_step(self, model, samples):
        return model(samples)["loss"]

    def valid_step(self, model, samples):
        raise NotImplementedError

    def before_evaluation(self, model, dataset, **kwargs):
        pass

    def after_evaluation(self, **kwargs):
        pass

    def evaluation(self, model, data_loader, cuda_enabled=True):
        metric_logger = MetricLogger(delimiter="  ")
        results = []
        for samples in metric_logger.log_every(data_loader, 10, "Evaluation"):
            results.extend(self.valid_step(model=model, samples=samples))
        if is_dist_avail_and_initialized():
            dist.barrier()
        return results

    @staticmethod
    def save_result(result, result_dir, filename, remove_duplicate=""):
        os.makedirs(result_dir, exist_ok=True)
        part = os.path.join(result_dir, "%s_rank%d.json" % (filename, get_rank()))
        final = os.path.join(result_dir, "%s.json" % filename)
        json.dump(result, open(part, "w"))
        if is_dist_avail_and_initialized():
            dist.barrier()
        if is_main_process():
            merged = []
            for r in range(get_world_size()):
                merged += json.load(open(os.path.join(result_dir, "%s_rank%d.json" % (filename, r))))
            if remove_duplicate:
                seen, uniq = set(), []
                for x in merged:
                    if x[remove_duplicate] not in seen:
                        seen.add(x[remove_duplicate])
                        uniq.append(x)
                merged = uniq
            json.dump(merged, open(final, "w"))
            logging.info("result file saved to %s" % final)
        return final
