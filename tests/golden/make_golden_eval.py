"""Golden for the moment-retrieval evaluator: the REFERENCE's lavis/tasks/mr_eval.py (eval_submission, compute_mr_ap, compute_mr_r1) run on
seeded synthetic submissions.  Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval.py -> mr_eval.json
(inputs + expected outputs; no reference source travels)."""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MRBLIP_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
for pkg in ("lavis", "lavis.tasks"):
    m = types.ModuleType(pkg)
    m.__path__ = [os.path.join(REF, *pkg.split("."))]
    sys.modules[pkg] = m
import importlib  # noqa: E402

ref = importlib.import_module("lavis.tasks.mr_eval")


def make_case(rng, n, kind):
    sub = []
    for q in range(n):
        dur = float(rng.integers(20, 151))
        ng = int(rng.integers(1, 4))
        gts = []
        for _ in range(ng):
            a = float(rng.integers(0, int(dur) - 2))
            gts.append([a, float(min(dur, a + rng.integers(2, 40)))])
        npred = int(rng.integers(1, 5)) if kind != "single" else 1
        preds = []
        for _ in range(npred):
            r = rng.random()
            if r < 0.35:      # jitter around a ground-truth window
                g = gts[int(rng.integers(0, ng))]
                preds.append([max(0.0, g[0] + float(rng.integers(-4, 5))), g[1] + float(rng.integers(-4, 5))])
            elif r < 0.5:     # exact hit
                preds.append(list(gts[int(rng.integers(0, ng))]))
            elif r < 0.6 and kind == "messy":
                preds.append([-1, -1])          # post_process's invalid marker
            elif r < 0.65 and kind == "messy":
                a = float(rng.integers(0, int(dur)))
                preds.append([a, a])            # zero-length window
            else:
                a = float(rng.integers(0, int(dur) - 1))
                preds.append([a, float(min(dur, a + rng.integers(1, 50)))])
        if kind == "messy" and rng.random() < 0.15:
            preds.append(list(preds[0]))        # duplicate prediction
        if kind == "int":
            gts = [[int(a), int(b)] for a, b in gts]
            preds = [[int(a), int(b)] for a, b in preds]
        sub.append({"qid": "q%d_%d" % (q, q % 3), "pred_relevant_windows": preds, "relevant_windows": gts})
    return sub


def clean(o):
    if isinstance(o, dict):
        return {str(k): clean(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [clean(v) for v in o]
    if isinstance(o, (np.floating, np.integer)):
        return o.item()
    return o


def main():
    rng = np.random.default_rng(20240917)
    cases = []
    for kind, n in (("single", 40), ("multi", 60), ("messy", 80), ("int", 30), ("multi", 1)):
        sub = make_case(rng, n, kind)
        full = ref.eval_submission(json.loads(json.dumps(sub)), json.loads(json.dumps(sub)), verbose=False)
        ap = ref.compute_mr_ap(json.loads(json.dumps(sub)), json.loads(json.dumps(sub)), num_workers=1)
        ap3 = ref.compute_mr_ap(json.loads(json.dumps(sub)), json.loads(json.dumps(sub)), num_workers=1, max_pred_windows=2, max_gt_windows=1)
        r1 = ref.compute_mr_r1(json.loads(json.dumps(sub)), json.loads(json.dumps(sub)))
        cases.append({"kind": kind, "submission": sub, "eval_submission": clean(full), "compute_mr_ap": clean(ap), "compute_mr_ap_capped": clean(ap3),
                      "compute_mr_r1": clean(list(r1))})
    json.dump({"cases": cases}, open(os.path.join(HERE, "mr_eval.json"), "w"))
    print("wrote mr_eval.json", [(c["kind"], len(c["submission"]), c["eval_submission"]["brief"]["MR-full-mAP"], c["eval_submission"]["brief"]["MR-full-R1-avg"]) for c in cases])


if __name__ == "__main__":
    main()
