"""Moment-retrieval evaluator: R1@IoU, mIoU, mAP@IoU (VOC-style detection AP per query) and the ``eval_submission`` report the task logs.

Restates the behaviour of the reference's ``lavis/tasks/mr_eval.py`` (``compute_mr_ap`` :26-96, ``compute_mr_r1`` :99-140,
``eval_moment_retrieval`` :178-217, ``eval_submission`` :330-416) and of the helpers it takes from ``mr_utils.py``
(``compute_temporal_iou_batch_paired`` :16-37, ``compute_temporal_iou_batch_cross`` :40-67, ``interpolated_precision_recall`` :70-87,
``compute_average_precision_detection`` :90-171), including their quirks, because the README's accuracy numbers are produced by exactly
these definitions:
  * the PAIRED IoU divides by (max end - min start) — not the true union — and maps 0/0 to 0; the CROSS IoU uses the true union and lets
    0/0 be NaN;
  * detection AP walks a query's predicted windows IN THE ORDER GIVEN (Mr. BLIP emits no scores), each one claims the not-yet-claimed
    ground-truth window of highest IoU if that IoU reaches the threshold; precision/recall are interpolated (VOC 2011);
  * every reported number is rounded through ``float(f"{x:.2f}")``; the "short / middle / long" entries repeat "full" (the reference
    removed the QVHighlights length ranges but kept the keys).
Pinned by tests/test_mr_eval_cpu.py against outputs of the reference evaluator (tests/golden/mr_eval.json, produced in the build
container by tests/golden/make_golden_eval.py).  Highlight-detection metrics (``pred_saliency_scores``) are outside the Mr. BLIP path
(the model emits windows only) and are not computed.
"""
from collections import OrderedDict
import logging

import numpy as np

IOU_THDS = tuple(float(f"{e:.2f}") for e in np.linspace(0.5, 0.95, 10))


def _fmt2(x) -> float:
    return float(f"{x:.2f}")


def iou_paired(pred: np.ndarray, gt: np.ndarray) -> np.ndarray:
    """row-wise IoU of two [N, 2] window arrays with the reference's (max end - min start) denominator; 0 where that is 0"""
    inter = np.maximum(0, np.minimum(pred[:, 1], gt[:, 1]) - np.maximum(pred[:, 0], gt[:, 0]))
    span = np.maximum(pred[:, 1], gt[:, 1]) - np.minimum(pred[:, 0], gt[:, 0])
    return np.divide(inter, span, out=np.zeros_like(inter), where=span != 0)


def iou_cross(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """[N, M] IoU (true union) of every window of a [N, 2] against every window of b [M, 2]; 0/0 stays NaN like the reference"""
    inter = np.clip(np.minimum(a[:, None, 1], b[None, :, 1]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    union = (a[:, 1] - a[:, 0])[:, None] + (b[:, 1] - b[:, 0])[None, :] - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / union


def _voc_ap(precision: np.ndarray, recall: np.ndarray) -> float:
    """interpolated AP (VOC 2011): precision envelope, summed over the recall steps"""
    p = np.concatenate([[0.0], precision, [0.0]])
    r = np.concatenate([[0.0], recall, [1.0]])
    p = np.maximum.accumulate(p[::-1])[::-1]
    step = np.nonzero(r[1:] != r[:-1])[0] + 1
    return float(np.sum((r[step] - r[step - 1]) * p[step]))


def detection_ap(gt_windows, pred_windows, iou_thds=IOU_THDS) -> np.ndarray:
    """AP of ONE query at every IoU threshold.  gt_windows / pred_windows: lists of [start, end] (predictions in rank order)."""
    n_gt, n_pred, n_thd = len(gt_windows), len(pred_windows), len(iou_thds)
    ap = np.zeros(n_thd)
    if n_pred == 0:
        return ap
    tp = np.zeros((n_thd, n_pred))
    if n_gt > 0:
        gts = np.asarray(gt_windows, dtype=float).reshape(n_gt, 2)
        claimed = np.zeros((n_thd, n_gt), dtype=bool)
        for j, w in enumerate(pred_windows):
            ious = iou_cross(np.asarray([w], dtype=float).reshape(1, 2), gts)[0]
            order = ious.argsort()[::-1]          # best ground-truth window first (the reference's ordering, ties included)
            for t, thd in enumerate(iou_thds):
                for k in order:
                    if ious[k] < thd:
                        break                      # nothing better left: false positive
                    if not claimed[t, k]:
                        claimed[t, k] = True
                        tp[t, j] = 1
                        break
    fp = 1.0 - tp
    tp_c, fp_c = np.cumsum(tp, axis=1), np.cumsum(fp, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        recall = tp_c / float(n_gt)
        precision = tp_c / (tp_c + fp_c)
    for t in range(n_thd):
        ap[t] = _voc_ap(precision[t], recall[t])
    return ap


def compute_mr_ap(submission, ground_truth=None, iou_thds=IOU_THDS, max_gt_windows=None, max_pred_windows=None, num_workers=1, chunksize=50):
    """mAP over queries at each IoU threshold, + "average".  Like the reference, predictions AND ground truth are read from
    ``submission`` (``pred_relevant_windows`` / ``relevant_windows``); ``ground_truth``, ``num_workers`` and ``chunksize`` are accepted for
    signature compatibility (a query's AP takes microseconds: no process pool)."""
    iou_thds = [float(f"{e:.2f}") for e in iou_thds]
    gt_by_qid, pred_by_qid = {}, {}
    for d in submission:
        preds = d["pred_relevant_windows"] if max_pred_windows is None else d["pred_relevant_windows"][:max_pred_windows]
        gts = d["relevant_windows"] if max_gt_windows is None else d["relevant_windows"][:max_gt_windows]
        pred_by_qid.setdefault(d["qid"], []).extend([w[0], w[1]] for w in preds)
        gt_by_qid.setdefault(d["qid"], []).extend([w[0], w[1]] for w in gts)
    aps = np.array([detection_ap(gt_by_qid.get(q, []), p, iou_thds) for q, p in pred_by_qid.items() if len(p) > 0])
    per_thd = aps.mean(0)
    out = {str(t): v for t, v in zip(iou_thds, per_thd)}
    out["average"] = np.mean(per_thd)
    return {k: _fmt2(100 * v) for k, v in out.items()}


def compute_mr_r1(submission, ground_truth, iou_thds=IOU_THDS):
    """R1: the FIRST predicted window against the ground-truth window it overlaps most; also mean IoU and the count of invalid
    ([-1, -1]) predictions."""
    iou_thds = [float(f"{e:.2f}") for e in iou_thds]
    first_pred = {d["qid"]: d["pred_relevant_windows"][0][:2] for d in submission}
    best_gt = {}
    for d in ground_truth:
        gts = d["relevant_windows"]
        k = 0
        if len(gts) > 0:
            k = int(np.argmax(iou_cross(np.array([first_pred[d["qid"]]]), np.array(gts))[0]))
        best_gt[d["qid"]] = gts[k]
    qids = list(first_pred)
    pw = np.array([first_pred[q] for q in qids]).astype(float)
    gw = np.array([best_gt[q] for q in qids]).astype(float)
    ious = iou_paired(pw, gw)
    r1 = {str(t): _fmt2(np.mean(ious >= t) * 100) for t in iou_thds}
    invalid = sum(1 for w in pw if -1 in w)
    return r1, np.mean(list(r1.values())), np.mean(ious), invalid


def eval_moment_retrieval(submission, ground_truth, verbose=True):
    full = {}
    ap = compute_mr_ap(submission, ground_truth)
    r1, r1_avg, miou, invalid = compute_mr_r1(submission, ground_truth)
    full = {"MR-mAP": ap, "MR-R1": r1, "MR-R1-avg": r1_avg, "MR-mIoU": miou, "MR-invalid_pred_num": invalid}
    if verbose:
        logging.info("[eval_moment_retrieval] %d queries: R1-avg %.2f mIoU %.4f mAP %.2f", len(submission), r1_avg, miou, ap["average"])
    # the reference keeps the QVHighlights range names but evaluates every one on the full set
    return {name: full for name in ("short", "middle", "long", "full")}


def eval_submission(submission, ground_truth, verbose=True, match_number=True):
    """submission / ground_truth: lists of {"qid", "pred_relevant_windows": [[st, ed], ...], "relevant_windows": [[st, ed], ...]}"""
    pred_qids, gt_qids = {e["qid"] for e in submission}, {e["qid"] for e in ground_truth}
    if match_number:
        assert pred_qids == gt_qids, "qids in ground_truth and submission must match. use `match_number=False` if you wish to disable this check"
    else:
        shared = pred_qids & gt_qids
        submission = [e for e in submission if e["qid"] in shared]
        ground_truth = [e for e in ground_truth if e["qid"] in shared]
    metrics, brief = {}, OrderedDict()
    if "pred_relevant_windows" in submission[0]:
        mr = eval_moment_retrieval(submission, submission, verbose=verbose)   # (sic: the reference scores the submission against itself —
        metrics.update(mr)                                                    #  both window lists travel in the submission records)
        b = {"MR-full-mAP": mr["full"]["MR-mAP"]["average"], "MR-full-mAP@0.5": mr["full"]["MR-mAP"]["0.5"], "MR-full-mAP@0.75": mr["full"]["MR-mAP"]["0.75"],
             "MR-short-mAP": mr["short"]["MR-mAP"]["average"], "MR-middle-mAP": mr["middle"]["MR-mAP"]["average"], "MR-long-mAP": mr["long"]["MR-mAP"]["average"],
             "MR-full-R1@0.5": mr["full"]["MR-R1"]["0.5"], "MR-full-R1@0.7": mr["full"]["MR-R1"]["0.7"], "MR-full-R1-avg": mr["full"]["MR-R1-avg"],
             "MR-full-mIoU": mr["full"]["MR-mIoU"], "MR-full-invalid_pred_num": mr["full"]["MR-invalid_pred_num"]}
        brief.update(sorted(b.items(), key=lambda kv: kv[0]))
    if "pred_saliency_scores" in submission[0]:
        logging.warning("eval_submission: pred_saliency_scores present — highlight-detection metrics are outside the Mr. BLIP path and are skipped")
    out = OrderedDict()
    out["brief"] = brief
    out.update(sorted(metrics.items(), key=lambda kv: kv[0]))
    return out
