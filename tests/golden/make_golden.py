"""Golden-vector generator: imports the REFERENCE (read-only, /root/reference) through ref_shim.py, loads
name-keyed synthetic weights (weights.py) into the reference's own modules at tiny dimensions, runs them
on CPU fp32 and stores inputs + expected outputs as small .npz fixtures next to this file.

Run in the build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
The fixtures are data (inputs / expected outputs / (key, shape) manifests); no reference source travels.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
from weights import seeded_state_dict  # noqa: E402
from mrblip.tokenizer import FixtureTokenizer  # noqa: E402

TINY = dict(
    vit=dict(embed_dim=96, depth=2, num_heads=4),
    bert=dict(hidden_size=64, num_attention_heads=4, intermediate_size=128, num_hidden_layers=4,
              vocab_size=100, max_position_embeddings=64),
    t5=dict(d_model=64, d_kv=16, d_ff=128, num_layers=2, num_decoder_layers=2, num_heads=4, vocab_size=32128,
            feed_forward_proj="gated-gelu", tie_word_embeddings=False, relative_attention_num_buckets=32,
            relative_attention_max_distance=128, dropout_rate=0.1, layer_norm_epsilon=1e-6),
)
IMG = 56
NQ = 8


def manifest_of(module):
    out = []
    for k, v in module.state_dict().items():
        if v.dtype in (torch.float32, torch.float64):
            out.append((k, list(v.shape)))
    return out


def load_seeded(module, prefix=""):
    """weights are keyed by the FULL reference parameter name (prefix + local name)."""
    man = [(prefix + k, s) for k, s in manifest_of(module)]
    sd = {k[len(prefix):]: v for k, v in seeded_state_dict(man).items()}
    missing = module.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    return man


def save(name, manifest=None, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    if manifest is not None:
        out["manifest_json"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in out.items()})


def main():
    torch.manual_seed(0)
    R = ref_shim.install(FixtureTokenizer, TINY)
    eva, qf, t5m, blip2, mr, mru, optims = R["eva"], R["qf"], R["t5m"], R["blip2"], R["mr"], R["mru"], R["optims"]
    g = torch.Generator().manual_seed(1234)

    # ---------------------------------------------------------------- ViT (eva_vit.py:324-340)
    vit = blip2.create_eva_vit_g(img_size=IMG, drop_path_rate=0.0, precision="fp32").eval()
    man = load_seeded(vit, "visual_encoder.")
    img = torch.randn(3, 3, IMG, IMG, generator=g)
    with torch.no_grad():
        y = vit(img)
        y0 = vit.blocks[0](vit.pos_drop(torch.cat((vit.cls_token.expand(3, -1, -1), vit.patch_embed(img)), 1) + vit.pos_embed))
    save("vit_tiny", man, image=img, out=y, block0=y0)

    # ---------------------------------------------------------------- ln_vision + Q-Former (Qformer.py:804-965)
    ln = blip2.LayerNorm(TINY["vit"]["embed_dim"])
    man_ln = load_seeded(ln, "ln_vision.")
    Q, qt = blip2.Blip2Base.init_Qformer(NQ, TINY["vit"]["embed_dim"])
    Q.cls = None
    Q.bert.embeddings.word_embeddings = None
    Q.bert.embeddings.position_embeddings = None
    for layer in Q.bert.encoder.layer:
        layer.output = None
        layer.intermediate = None
    Q.eval()
    man_q = load_seeded(Q, "Qformer.")
    qt_val = seeded_state_dict([("query_tokens", list(qt.shape))])["query_tokens"]
    with torch.no_grad():
        emb = ln(y)
        atts = torch.ones(emb.shape[:-1], dtype=torch.long)
        o = Q.bert(query_embeds=qt_val.expand(3, -1, -1), encoder_hidden_states=emb,
                   encoder_attention_mask=atts, return_dict=True).last_hidden_state
    save("qformer_tiny", man_ln + man_q + [("query_tokens", list(qt.shape))], vit_out=y, ln_out=emb, out=o)

    # ---------------------------------------------------------------- T5 (modeling_t5.py:1734-1893)
    cfg = R["t5_config"]()
    cfg.dense_act_fn = "gelu"
    t5 = t5m.T5ForConditionalGeneration(cfg).eval()
    man_t5 = load_seeded(t5, "t5_model.")
    B, S, Ld = 2, 150, 7
    x = torch.randn(B, S, cfg.d_model, generator=g) * 0.5
    am = torch.ones(B, S, dtype=torch.long)
    am[1, 140:] = 0
    labels = torch.randint(5, 1000, (B, Ld), generator=g)
    labels[1, 5:] = -100
    dm = (labels != -100).long()
    xr = x.clone().requires_grad_(True)
    o = t5(inputs_embeds=xr, attention_mask=am, decoder_attention_mask=dm, labels=labels, return_dict=True)
    o.loss.backward()
    save("t5_tiny", man_t5, inputs_embeds=x, attention_mask=am, labels=labels, loss=o.loss, logits_sub=o.logits[..., ::64], logits_lse=torch.logsumexp(o.logits, -1),
         enc_out=o.encoder_last_hidden_state, d_inputs_embeds=xr.grad,
         shift_right=t5._shift_right(labels))

    # integer goldens: relative position buckets (modeling_t5.py:392-445)
    rel = torch.arange(-300, 301)[None, :]
    save("t5_buckets",
         rel=rel[0],
         bidir=t5m.T5Attention._relative_position_bucket(rel, True, 32, 128)[0],
         unidir=t5m.T5Attention._relative_position_bucket(rel, False, 32, 128)[0])

    # ---------------------------------------------------------------- full BLIP2_MR.forward_mr (blip2_mr.py:433-570)
    for tag, agg in (("mr_tiny", False), ("mr_tiny_mean", "mean")):
        torch.manual_seed(0)
        model = mr.BLIP2_MR(img_size=IMG, vit_precision="fp32", num_query_token=NQ, t5_model="google/flan-t5-xl",
                            input_time_format="seconds_integers", interleave_data=True,
                            frame_token_aggregation=agg, task="qformer_freeze_lora")
        model.eval()
        man = load_seeded(model)
        Bv, T = 2, 3
        video = torch.randn(Bv, T, 3, IMG, IMG, generator=g)
        timestamps = torch.tensor([[2.5, 22.49, 39.0], [7.0, 105.2, 187.6]], dtype=torch.float32)
        duration = torch.tensor([45.0, 250.0])
        samples = dict(
            video=video, timestamps=timestamps, duration=duration,
            query_prompt=["Query: a man opens the red door\n", "Query: the dog runs\n"],
            task_prompt=["Given the video and the query, find the relevant windows.\nRelevant windows: "] * 2,
            video_prompt_end=["<extra_id_0>"] * 2,
            relevant_windows=["[[8, 16]]", "[[0, 4], [22, 150]]"],
        )
        cap = {}
        orig_pc = model.prompt_concatenation

        def pc(*a, **k):
            r = orig_pc(*a, **k)
            cap["embs"], cap["atts"], cap["prompt"] = r[0].detach().clone(), r[1].clone(), r[2]
            return r

        model.prompt_concatenation = pc
        orig_t5_forward = model.t5_model.forward

        def t5f(*a, **k):
            r = orig_t5_forward(*a, **k)
            cap["logits"] = r.logits.detach().clone()
            cap["labels"] = k["labels"].clone()
            return r

        model.t5_model.forward = t5f
        out = model(samples)
        out["loss"].backward()
        trainable = sorted(n for n, p in model.named_parameters() if p.requires_grad and p.grad is not None)
        top = sorted({n.split(".")[0] for n, p in model.named_parameters() if p.requires_grad})
        grads = {"grad__" + n.replace(".", "__"): dict(model.named_parameters())[n].grad
                 for n in ["t5_proj.weight", "t5_proj.bias", "ln_vision.weight", "ln_vision.bias"]}
        save(tag, man, video=video, timestamps=timestamps, duration=duration, loss=out["loss"],
             inputs_embs=cap["embs"], inputs_atts=cap["atts"], logits_sub=cap["logits"][..., ::64], logits_lse=torch.logsumexp(cap["logits"], -1), labels=cap["labels"],
             annoying=np.array(model.annoying_numbers),
             annoying_map=np.array(sorted(model.annoying_numbers_replacement_dict.items())),
             strings_json=np.frombuffer(json.dumps(dict(
                 query_prompt=samples["query_prompt"], task_prompt=samples["task_prompt"],
                 video_prompt_end=samples["video_prompt_end"], relevant_windows=samples["relevant_windows"],
                 video_prompt=cap["prompt"], trainable_top=top, n_trainable=len(trainable))).encode(), dtype=np.uint8),
             **grads)

    # ---------------------------------------------------------------- timestamp formatting (utils.py:388-434)
    amap = model.annoying_numbers_replacement_dict
    ts = torch.tensor([[0.5, 1.5, 2.5, 3.5, 4.49, 5.0, 21.7, 22.2, 38.6, 7.4, 56.0, 198.9]], dtype=torch.float32)
    nt, nd, vp = mru.get_timestamps_as_seconds_integers(ts, torch.tensor([39.4]), amap)
    save("timestamps", ts=ts, dur=np.array([39.4], dtype=np.float32), out=nt[0], out_dur=np.array(nd),
         prompt_json=np.frombuffer(json.dumps(vp).encode(), dtype=np.uint8))

    # ---------------------------------------------------------------- post_process / moment_str_to_list (utils.py:18-83, 300-341)
    cases = ["[[8, 16]]", "[[8, 16], [20, 30]]", "[[8, 16], [20, 30]", "[8, 16]]", "[[8, 16", "[[8.5, 16.25]]",
             "[[8, 16]] extra", "[[8,16],[3,4]]", "8, 16", "[[16, 8]]", "", "[[1, 2], [3, 4], [5, 6]]", "[[ 8 , 16 ]]",
             "[[8, 16]][[1, 2]]", "[[8 16]]", "[[a, b]]"]
    pp, ml = [], []
    for c in cases:
        try:
            p = mru.post_process(c)
        except Exception as e:  # noqa
            p = "EXC:" + type(e).__name__
        pp.append(p)
        try:
            m = mru.moment_str_to_list(p if not str(p).startswith("EXC:") else c)
        except Exception as e:  # noqa
            m = "EXC:" + type(e).__name__
        ml.append(m)
    save("post_process", cases_json=np.frombuffer(json.dumps(dict(cases=cases, post=pp, moments=ml)).encode(), dtype=np.uint8))

    # ---------------------------------------------------------------- LR schedule (optims.py:56-119)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sched = optims.LinearWarmupCosineLRScheduler(opt, max_epoch=5, min_lr=0.0, init_lr=3e-4, warmup_steps=30,
                                                 warmup_start_lr=1e-8, iters_per_epoch=20)
    lrs = []
    for ep in range(5):
        for it in range(20):
            sched.step(cur_epoch=ep, cur_step=ep * 20 + it)
            lrs.append(opt.param_groups[0]["lr"])
    save("lr_sched", lrs=np.array(lrs, dtype=np.float64))


if __name__ == "__main__":
    main()
