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
from lavis.tasks.mr_eval import eval_submission


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
        """R1@IoU, mAP@IoU, mIoU, invalid-prediction rate and agg_metrics = R1 averaged over IoU 0.5:0.05:0.95, computed by the
        evaluator of lavis/tasks/mr_eval.py exactly as the reference does (moment_retrieval.py:115-152)."""
        if not is_main_process():
            return None
        results = json.load(open(eval_result_file))
        total = len(results)
        interpreted = [{"qid": r["qid"], "pred_relevant_windows": moment_str_to_list(r["prediction"]),
                        "relevant_windows": moment_str_to_list(str(r["target"]))} for r in results]
        allm = eval_submission(interpreted, interpreted, verbose=False)
        metrics = {"agg_metrics": allm["brief"]["MR-full-R1-avg"], "r1": allm["full"]["MR-R1"], "mAP": allm["full"]["MR-mAP"],
                   "mIoU": allm["brief"]["MR-full-mIoU"], "invalid_predictions": allm["brief"]["MR-full-invalid_pred_num"] / max(total, 1),
                   "total": total}
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
        pending_loss = None  # the loss is read ONE step late: loss.item() right after the step (the reference's loop, :234) drains the
        #                      stream, and with ~2900 launches per step the host would then re-fill the queue while the GPU idles
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
            if pending_loss is not None:
                metric_logger.update(loss=pending_loss.item())
            pending_loss = loss.detach()
            metric_logger.update(lr=optimizer.lr)
        if pending_loss is not None:
            metric_logger.update(loss=pending_loss.item())
        metric_logger.synchronize_between_processes()
        logging.info("Averaged stats: " + metric_logger.global_avg())
        return {k: "{:.3f}".format(m.global_avg) for k, m in metric_logger.meters.items()}
