"""Golden vectors at the BENCHED size (BASELINE.json configs[1], "C2": the QVHighlights fine-tuning shape): the reference's own
``BLIP2_MR.forward_mr`` + ``loss.backward()`` with the full ViT-g/14 (39 blocks), the bert-base-sized Q-Former (12 layers, 32 queries)
and a Flan-T5-XL-sized T5 (24 + 24 layers, d 2048, 32 heads of 64, d_ff 5120), 60 frames of 224x224, batch 1, CPU fp32, eval mode.

Run in the build container only (imports /root/reference through ref_shim.py; ~45 GB RAM, ~15 minutes on 8 cores):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_c2.py
Writes tests/golden/mr_c2.npz: inputs (timestamps, strings; the frames are regenerated from a seed), the (key, shape) manifest of the
reference's state dict (weights are regenerated from the key names by weights.py, wscale=0.25 / fast=True), and SUB-SAMPLED expected
outputs of every tower + loss + logits + the gradients of the trainable non-LoRA tensors.  No reference source travels.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
from weights import seeded_array  # noqa: E402
from mrblip.tokenizer import FixtureTokenizer  # noqa: E402

C2 = dict(
    vit=dict(embed_dim=1408, depth=39, num_heads=16),
    bert=dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, num_hidden_layers=12, vocab_size=100,
              max_position_embeddings=64),
    t5=dict(d_model=2048, d_kv=64, d_ff=5120, num_layers=24, num_decoder_layers=24, num_heads=32, vocab_size=32128,
            feed_forward_proj="gated-gelu", tie_word_embeddings=False, relative_attention_num_buckets=32,
            relative_attention_max_distance=128, dropout_rate=0.1, layer_norm_epsilon=1e-6),
)
WSCALE = 0.15   # generic Linear weights ~ N(0, 0.012): T5 attention is unscaled (no 1/sqrt(d_kv)), so at d_model 2048 a 0.02 std gives score std ~6.5 over 2012 keys —
                # near-one-hot softmaxes that amplify fp32 round-off 40x through the decoder (measured: oracle-fp32 vs reference-fp32 logits 4.4e-3);
                # 0.012 keeps the score std ~2.4, the conditioning the C1 fixture has at d_model 768
T = 60
DURATION = 150.0


def c2_samples():
    """the bench's QVH clip shape with its own prompt / answer (bench.py: synthetic_samples) — frames drawn from a seed"""
    video = torch.from_numpy(seeded_array("c2.input.video", (1, T, 3, 224, 224), std=1.0, fast=True))
    ts = torch.tensor([[round((i + 0.5) * DURATION / T, 2) for i in range(T)]], dtype=torch.float32)
    return dict(video=video, timestamps=ts, duration=torch.tensor([DURATION]),
                query_prompt=["Query: a person opens the red door and walks into the kitchen\n"],
                task_prompt=["Given the video and the query, find the relevant windows.\nRelevant windows: "],
                video_prompt_end=["<extra_id_0>"], relevant_windows=["[[8, 16], [92, 110]]"])


def main():
    t0 = time.time()
    torch.manual_seed(0)
    R = ref_shim.install(FixtureTokenizer, C2)
    mr = R["mr"]
    model = mr.BLIP2_MR(img_size=224, vit_precision="fp32", num_query_token=32, t5_model="google/flan-t5-xl",
                        input_time_format="seconds_integers", interleave_data=True, frame_token_aggregation=False,
                        task="qformer_freeze_lora")
    model.eval()
    man = [(k, list(v.shape)) for k, v in model.state_dict().items() if v.dtype in (torch.float32, torch.float64)]
    print("built reference model: %d tensors, %.1f M params, %.0f s" % (len(man), sum(np.prod(s) for _, s in man) / 1e6, time.time() - t0), flush=True)
    with torch.no_grad():   # tensor by tensor (a second 16 GB copy of the state dict would not fit beside the activations)
        own = model.state_dict()
        for k, s in man:
            own[k].copy_(torch.from_numpy(seeded_array(k, s, wscale=WSCALE, fast=True)))
        del own
    print("weights loaded %.0f s" % (time.time() - t0), flush=True)
    samples = c2_samples()
    cap = {}
    model.visual_encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("vit", o.detach()[::6, ::16, ::16].clone()))
    model.ln_vision.register_forward_hook(lambda m, i, o: cap.__setitem__("ln", o.detach()[::6, ::16, ::16].clone()))
    model.Qformer.bert.register_forward_hook(lambda m, i, o: cap.__setitem__("qf", o.last_hidden_state.detach()[::6, ::4, ::8].clone()))
    orig_pc = model.prompt_concatenation

    def pc(*a, **k):
        r = orig_pc(*a, **k)
        cap["embs"], cap["atts"], cap["prompt"] = r[0].detach()[:, ::4, ::16].clone(), r[1].clone(), r[2]
        return r

    model.prompt_concatenation = pc
    orig_t5_forward = model.t5_model.forward

    def t5f(*a, **k):
        r = orig_t5_forward(*a, **k)
        cap["logits"] = r.logits.detach().clone()
        cap["labels"] = k["labels"].clone()
        cap["enc"] = r.encoder_last_hidden_state.detach()[:, ::4, ::16].clone()
        return r

    model.t5_model.forward = t5f
    t1 = time.time()
    out = model(samples)
    t2 = time.time()
    print("reference forward %.1f s, loss %.5f" % (t2 - t1, out["loss"].item()), flush=True)
    out["loss"].backward()
    t3 = time.time()
    print("reference backward %.1f s" % (t3 - t2), flush=True)
    named = dict(model.named_parameters())
    grads = {"grad__" + n.replace(".", "__"): named[n].grad for n in ["t5_proj.weight", "t5_proj.bias", "ln_vision.weight", "ln_vision.bias"]}
    grads["grad__t5_proj__weight"] = grads["grad__t5_proj__weight"][::16, ::4]
    arrs = dict(
        timestamps=samples["timestamps"], duration=samples["duration"], loss=out["loss"].detach(),
        vit_sub=cap["vit"], ln_sub=cap["ln"], qf_sub=cap["qf"], inputs_embs_sub=cap["embs"], inputs_atts=cap["atts"], enc_sub=cap["enc"],
        logits_sub=cap["logits"][..., ::64], logits_lse=torch.logsumexp(cap["logits"], -1), labels=cap["labels"], **grads,
    )
    outd = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        outd[k] = np.asarray(v)
    outd["manifest_json"] = np.frombuffer(json.dumps(man).encode(), dtype=np.uint8)
    outd["strings_json"] = np.frombuffer(json.dumps(dict(
        query_prompt=samples["query_prompt"], task_prompt=samples["task_prompt"], video_prompt_end=samples["video_prompt_end"],
        relevant_windows=samples["relevant_windows"], video_prompt=cap["prompt"], wscale=WSCALE, T=T,
        subsample=dict(vit="[::6, ::16, ::16]", ln="[::6, ::16, ::16]", qf="[::6, ::4, ::8]", inputs_embs="[:, ::4, ::16]", enc="[:, ::4, ::16]",
                       logits="[..., ::64]", t5_proj_weight_grad="[::16, ::4]"),
        ref_forward_s=t2 - t1, ref_backward_s=t3 - t2, ref_threads=torch.get_num_threads())).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "mr_c2.npz"), **outd)
    print("wrote mr_c2.npz", {k: getattr(v, "shape", None) for k, v in outd.items()}, "%.0f s total" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
