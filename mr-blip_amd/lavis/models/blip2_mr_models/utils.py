"""Text post-processing of generated windows (behaviour of lavis/models/blip2_mr_models/utils.py:18-83, 300-341)."""
import ast
import re


def post_process(pred: str) -> str:
    """repair a predicted "[[s, e], ...]" string: missing commas, duplicate commas, swapped start/end; "[[-1, -1]]" if hopeless."""
    pred = pred.split("</s>")[0]
    if not re.match(r"\[\[.*\]\]", pred):
        return "[[-1, -1]]"
    body = pred[1:-1]
    fixed = []
    for win in re.split(r"\s+(?=\[)", body):
        win = re.sub(r",+$", "", win)
        win = re.sub(r"(\d) (\d)", r"\1, \2", win)
        win = re.sub(r",+", ",", win)
        nums = re.findall(r"\d+", win)
        if len(nums) == 2 and int(nums[0]) > int(nums[1]):
            win = "[" + nums[1] + ", " + nums[0] + "]"
        fixed.append(win)
    return "[" + ", ".join(fixed) + "]"


def moment_str_to_list(m: str):
    if m == "[[-1, -1]]" or not re.match(r"\[\[.*\]\]", m):
        return [[-1, -1]]
    try:
        val = ast.literal_eval(m)
    except Exception:
        return [[-1, -1]]
    if not isinstance(val, list):
        return [[-1, -1]]
    return [w if len(w) == 2 else [-1, -1] for w in val]
