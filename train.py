#!/usr/bin/env python
"""Entry point kept from the reference (train.py:38-130):
    python -m torch.distributed.run --nproc_per_node=N train.py --cfg-path mr-blip_amd/lavis/projects/mr_BLIP/train/qvh.yaml [--options k=v ...]
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import lavis.tasks as tasks  # noqa: E402
from lavis.common.config import Config  # noqa: E402
from lavis.common.dist_utils import get_rank, init_distributed_mode  # noqa: E402
from lavis.common.logger import setup_logger  # noqa: E402
from lavis.common.registry import registry  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Training")
    p.add_argument("--cfg-path", required=True, help="path to configuration file.")
    p.add_argument("--options", nargs="+", help="override settings in xxx=yyy format")
    return p.parse_args(argv)


def setup_seeds(config):
    seed = config.run_cfg.seed + get_rank()
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def main(argv=None, evaluate=False):
    job_id = time.strftime("%Y%m%d%H%M")[:-1]
    cfg = Config(parse_args(argv))
    init_distributed_mode(cfg.run_cfg)
    setup_seeds(cfg)
    setup_logger()
    cfg.pretty_print()
    task = tasks.setup_task(cfg)
    datasets = task.build_datasets(cfg)
    model = task.build_model(cfg)
    runner = registry.get_runner_class(cfg.run_cfg.get("runner", "runner_base"))(cfg=cfg, job_id=job_id, task=task, model=model, datasets=datasets)
    if evaluate:
        return runner.evaluate(skip_reload=True)
    return runner.train()


if __name__ == "__main__":
    main()
