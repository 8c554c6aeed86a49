"""f1 — the evaluator (lavis/tasks/mr_eval.py of this package) against outputs of the REFERENCE evaluator on seeded synthetic
submissions (tests/golden/mr_eval.json <- tests/golden/make_golden_eval.py): eval_submission incl. mAP per IoU threshold, R1, mIoU, invalid
count; compute_mr_ap with window caps; plus the task's _report_metrics on a result file in the reference's format."""
import json
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))


def _same(a, b, path="", ap_slack=False):
    """ap_slack: inside eval_submission the reference gathers per-query APs from an 8-process pool in completion order
    (pool.imap_unordered, mr_eval.py:80-86), so its mean over queries is summed in a run-dependent order and a value sitting exactly on
    a rounding boundary (e.g. 29.375) prints as either neighbour; the single-process goldens (compute_mr_ap) are compared exactly."""
    if ap_slack and isinstance(b, float) and "mAP" in path:
        assert abs(float(a) - b) <= 0.0100001, (path, a, b)
        return
    if isinstance(b, dict):
        assert isinstance(a, dict) and set(map(str, a)) == set(b), (path, sorted(map(str, a)), sorted(b))
        for k in b:
            _same({str(x): y for x, y in a.items()}[k], b[k], path + "/" + k, ap_slack)
    elif isinstance(b, list):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, path + "[%d]" % i, ap_slack)
    elif isinstance(b, float):
        assert (math.isnan(a) and math.isnan(b)) or abs(float(a) - b) <= 1e-9 * max(1.0, abs(b)), (path, a, b)
    else:
        assert a == b, (path, a, b)


def test_eval_submission_matches_reference():
    from lavis.tasks import mr_eval as E

    g = json.load(open(os.path.join(ROOT, "tests", "golden", "mr_eval.json")))
    assert len(g["cases"]) == 5
    for c in g["cases"]:
        sub = c["submission"]
        _same(json.loads(json.dumps(E.eval_submission(sub, sub, verbose=False))), c["eval_submission"], c["kind"], ap_slack=True)
        _same(E.compute_mr_ap(sub, sub), c["compute_mr_ap"], c["kind"] + ":ap")
        _same(E.compute_mr_ap(sub, sub, max_pred_windows=2, max_gt_windows=1), c["compute_mr_ap_capped"], c["kind"] + ":ap_capped")
        r1, r1_avg, miou, invalid = E.compute_mr_r1(sub, sub)
        _same([r1, float(r1_avg), float(miou), int(invalid)], c["compute_mr_r1"], c["kind"] + ":r1")


def test_eval_submission_qid_mismatch():
    from lavis.tasks import mr_eval as E

    a = [{"qid": "a", "pred_relevant_windows": [[0, 4]], "relevant_windows": [[0, 5]]}]
    b = [{"qid": "b", "pred_relevant_windows": [[0, 4]], "relevant_windows": [[0, 5]]}]
    with pytest.raises(AssertionError):
        E.eval_submission(a, b)
    out = E.eval_submission(a + b, a, match_number=False, verbose=False)
    assert out["brief"]["MR-full-R1@0.7"] == 100.0 and out["brief"]["MR-full-mAP@0.75"] == 100.0 and out["brief"]["MR-full-mAP"] == 70.0


def test_task_report_metrics_uses_the_reference_definitions(tmp_path):
    """moment_retrieval._report_metrics: prediction / target STRINGS -> windows (moment_str_to_list) -> eval_submission ->
    {agg_metrics (= R1 averaged over IoU 0.5:0.05:0.95), r1, mAP, mIoU, invalid_predictions, total} (moment_retrieval.py:115-152)."""
    import lavis  # noqa: F401
    from lavis.tasks import mr_eval as E
    from lavis.tasks.moment_retrieval import MomentRetrievalTask

    res = [{"qid": "1_0", "prediction": "[[8, 16]]", "raw_prediction": "[[8, 16]]", "target": "[[8, 16]]", "duration": 150.0},
           {"qid": "2_1", "prediction": "[[0, 10], [40, 60]]", "raw_prediction": "x", "target": "[[2, 10], [30, 60]]", "duration": 150.0},
           {"qid": "3_2", "prediction": "[[-1, -1]]", "raw_prediction": "garbage", "target": "[[5, 9]]", "duration": 30.0}]
    f = tmp_path / "val_epoch0.json"
    json.dump(res, open(f, "w"))
    m = MomentRetrievalTask()._report_metrics(str(f), "val")
    assert set(m) == {"agg_metrics", "r1", "mAP", "mIoU", "invalid_predictions", "total"}
    assert m["total"] == 3 and abs(m["invalid_predictions"] - 1 / 3) < 1e-12
    sub = [{"qid": r["qid"], "pred_relevant_windows": w, "relevant_windows": t} for r, w, t in
           zip(res, ([[8, 16]], [[0, 10], [40, 60]], [[-1, -1]]), ([[8, 16]], [[2, 10], [30, 60]], [[5, 9]]))]
    ref = E.eval_submission(sub, sub, verbose=False)
    assert m["agg_metrics"] == ref["brief"]["MR-full-R1-avg"] and m["mAP"] == ref["full"]["MR-mAP"] and m["r1"] == ref["full"]["MR-R1"]
    assert m["r1"]["0.5"] == 66.67 and m["mAP"]["0.5"] == 66.67     # clip 1 exact, clip 2's first window IoU 0.8 with [2, 10], clip 3 invalid
