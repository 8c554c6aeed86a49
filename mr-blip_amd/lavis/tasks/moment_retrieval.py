"""``moment_retrieval`` task: the micro-step loop with gradient accumulation and the generate-based evaluation
(behaviour of lavis/tasks/moment_retrieval.py:33-257).  Differences, both documented in DESIGN.md: bf16 needs no GradScaler, and
the gradient all-reduce happens ONCE per optimizer step (accumulation boundary), not per micro-step."""
import json
import logging

import numpy as np
import torch

from lavis.common.dist_utils import is_main_process
from lavis.common.logger import MetricLogger
from lavis.common.registry import registry
from lavis.models.blip2_mr_models.utils import moment_str_to_list
from lavis.tasks.base_task import BaseTask


def temporal_iou(a, b):
    inter = max(0.0, min(a[1], b[1]) - max(a[0], b[0]))
    union = max(a[1], b[1]) - min(a[0], b[0])
    return inter / union if union > 0 else 0.0


def _with_next(it):
    """(item, next item or None) pairs of an iterable"""
    it = iter(it)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


@registry.register_task("moment_retrieval")
class MomentRetrievalTask(BaseTask):
    def valid_step(self, model, samples):
        out = model.generate(samples)
        res = []
        for i, (a, q, p, rp, d) in enumerate(zip(out["answer"], out["qid"], out["prediction"], out["raw_prediction"], out["duration"])):
            res.append({"qid": str(q) + "_" + str(i), "raw_prediction": rp, "prediction": p, "target": a, "duration": d})
        return res

    def after_evaluation(self, val_result, split_name, epoch, **kwargs):
        f = self.save_result(val_result, registry.get_path("result_dir") or "result", "{}_epoch{}".format(split_name, epoch))
        return self._report_metrics(f, split_name)

    def _report_metrics(self, eval_result_file, split_name):
        """R1@IoU{0.5,0.7}, mIoU and agg_metrics = mean R1 over IoU 0.5:0.05:0.95 of the top-1 window (moment_retrieval.py:115-152)."""
        if not is_main_process():
            return None
        results = json.load(open(eval_result_file))
        thresholds = np.arange(0.5, 0.96, 0.05)
        ious, invalid = [], 0
        for r in results:
            pred = moment_str_to_list(r["prediction"])
            gt = moment_str_to_list(str(r["target"]))
            if pred == [[-1, -1]]:
                invalid += 1
                ious.append(0.0)
                continue
            ious.append(max(temporal_iou(pred[0], g) for g in gt))
        ious = np.array(ious) if ious else np.zeros(1)
        r1 = {float(round(t, 2)): float((ious >= t).mean() * 100) for t in thresholds}
        metrics = {"r1": r1, "mIoU": float(ious.mean() * 100), "invalid_predictions": invalid / max(len(results), 1), "total": len(results),
                   "agg_metrics": float(np.mean(list(r1.values())))}
        logging.info(metrics)
        return metrics

    def train_epoch(self, epoch, model, data_loader, optimizer, lr_scheduler, log_freq=50, accum_grad_iters=1, reduce_grads=None,
                    arm_exchange=None, **kwargs):
        """one epoch: lr step -> forward/backward (HIP) -> every accum_grad_iters: all-reduce grads once, AdamW, zero_grad.  Like the
        reference (moment_retrieval.py:205-232): loss.backward() is NOT divided by accum_grad_iters, and there is NO zero_grad at the
        start of an epoch — micro-steps left over when len(loader) % accum_grad_iters != 0 carry into the next epoch's first step."""
        metric_logger = MetricLogger(delimiter="  ")
        iters_per_epoch = len(data_loader)
        model.train()
        for i, (samples, nxt) in enumerate(_with_next(metric_logger.log_every(data_loader, log_freq, f"Train: data epoch: [{epoch}]"))):
            samples.update({"epoch": epoch, "num_iters_per_epoch": iters_per_epoch, "iters": i})
            if nxt is not None:  # one-batch look-ahead: the model overlaps the next clip's frozen-ViT forward with this step's decoder
                samples["next_video"] = nxt["video"]
            lr_scheduler.step(cur_epoch=epoch, cur_step=i)
            if arm_exchange is not None and (i + 1) % accum_grad_iters == 0:
                arm_exchange()  # this micro-step closes the window: its backward starts the (overlapped) gradient all-reduce
            loss = self.train_step(model=model, samples=samples)
            loss.backward()
            if (i + 1) % accum_grad_iters == 0:
                if reduce_grads is not None:
                    reduce_grads()
                optimizer.step()
                optimizer.zero_grad()
            metric_logger.update(loss=loss.item(), lr=optimizer.lr)
        metric_logger.synchronize_between_processes()
        logging.info("Averaged stats: " + metric_logger.global_avg())
        return {k: "{:.3f}".format(m.global_avg) for k, m in metric_logger.meters.items()}
