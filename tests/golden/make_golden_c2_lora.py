"""One-off (build container, ~50 GB RAM, ~10 minutes): the fp32 ORACLE at the benched size C2 with NON-ZERO LoRA — every adapter's dA / dB at
Flan-T5-XL width (d 2048, d_ff 5120, 24 + 24 layers, S = 2012), sub-sampled, for tests/test_fullsize_gpu.py::
test_c2_benched_size_nonzero_lora_gradients.  VERDICT r4 "missing 2": the benched model's LoRA path (the K = 10240 dX, the stacked cross K / V,
the thin-role kernels that only engage at >= 512 rows) had been compared with nothing but itself.

The REFERENCE cannot provide this fixture: peft is absent from the image and the reference-side harness stubs `get_peft_model` as identity
(SURVEY.md section 8c), so the vectors come from the ORACLE's restatement of peft's published algorithm y = W x + (alpha / r) B A dropout(x)
(blip2_mr.py:182-200, 236) — "parity unpinned" for the LoRA numerics, as DESIGN.md section 2 says; what IS pinned is the oracle's LoRA-free
path at this very size (check_oracle_c2.py: loss bit-equal, gradients 1e-6 against the reference-generated mr_c2.npz).
    python tests/golden/make_golden_c2_lora.py          ->  tests/golden/mr_c2_lora.npz
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "mr-blip_amd"), ROOT, os.path.join(ROOT, "tests"), HERE):
    sys.path.insert(0, p)

from util import load_golden  # noqa: E402
from weights import seeded_array  # noqa: E402
from mrblip import prompt as P  # noqa: E402
from mrblip.tokenizer import FixtureTokenizer  # noqa: E402
from oracle import mrblip_oracle as O  # noqa: E402
from check_oracle_c2 import CFG  # noqa: E402
from test_model_gpu import _peft_sd  # noqa: E402

LORA_STD = 0.02      # the bench's lora_init_nonzero scale
STRIDE = 32          # sub-sampling of dA columns / dB rows


def main():
    g = load_golden("mr_c2")
    st = g["strings"]
    t0 = time.time()
    sd = {k: torch.from_numpy(seeded_array(k, s, wscale=st["wscale"], fast=True)) for k, s in g["manifest"]}
    sdl = _peft_sd(sd, lora_std=LORA_STD)
    del sd
    train_keys = [k for k in sdl if ("lora_" in k) or k.startswith("t5_proj") or k.startswith("ln_vision")]
    for k in train_keys:
        sdl[k].requires_grad_(True)
    T = int(st["T"])
    samples = dict(video=torch.from_numpy(seeded_array("c2.input.video", (1, T, 3, 224, 224), std=1.0, fast=True)),
                   timestamps=torch.from_numpy(g["timestamps"]), duration=torch.from_numpy(g["duration"]), query_prompt=st["query_prompt"],
                   task_prompt=st["task_prompt"], video_prompt_end=st["video_prompt_end"], relevant_windows=st["relevant_windows"])
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    orc = O.Oracle(sdl, CFG, lora=dict(r=8, alpha=8))
    t1 = time.time()
    out = orc.forward_mr(tok, samples, repl)
    t2 = time.time()
    out["loss"].backward()
    t3 = time.time()
    print("oracle C2 + LoRA: weights %.0f s, forward %.1f s, backward %.1f s, loss %.6f" % (t1 - t0, t2 - t1, t3 - t2, out["loss"].item()))
    assert np.array_equal(out["inputs_atts"].numpy(), g["inputs_atts"]) and np.array_equal(out["labels"].numpy(), g["labels"])
    arrays = dict(loss=np.float64(out["loss"].item()),
                  enc_sub=out["enc"].detach()[:, ::4, ::16].numpy(), logits_sub=out["logits"].detach()[..., ::64].numpy(),
                  logits_lse=torch.logsumexp(out["logits"].detach(), -1).numpy(),
                  grad__t5_proj__weight=sdl["t5_proj.weight"].grad[::16, ::4].numpy(), grad__t5_proj__bias=sdl["t5_proj.bias"].grad.numpy(),
                  grad__ln_vision__weight=sdl["ln_vision.weight"].grad.numpy(), grad__ln_vision__bias=sdl["ln_vision.bias"].grad.numpy())
    names, norms = [], []
    da_parts, db_parts = [], []
    for k in sorted(sdl):
        if not k.endswith(".lora_A.default.weight"):
            continue
        base = k[: -len(".lora_A.default.weight")]
        ga, gb = sdl[k].grad, sdl[base + ".lora_B.default.weight"].grad
        names.append(base[len("t5_model.base_model.model."):])
        norms.append([float(ga.double().pow(2).sum().sqrt()), float(gb.double().pow(2).sum().sqrt())])
        da_parts.append(ga[:, ::STRIDE].reshape(-1).numpy())
        db_parts.append(gb[::STRIDE].reshape(-1).numpy())
    arrays["lora_dA_sub"] = np.concatenate(da_parts).astype(np.float32)     # per adapter: [8, in / STRIDE] row-major, adapters in `names` order
    arrays["lora_dB_sub"] = np.concatenate(db_parts).astype(np.float32)     # per adapter: [out / STRIDE, 8]
    arrays["lora_norms"] = np.asarray(norms, dtype=np.float64)              # [n_adapters, 2]: full Frobenius norms of dA, dB
    arrays["strings_json"] = np.frombuffer(json.dumps(dict(names=names, stride=STRIDE, lora_std=LORA_STD)).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "mr_c2_lora.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KB;", len(names), "adapters")


if __name__ == "__main__":
    main()
