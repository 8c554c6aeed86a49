"""f3 — checkpoint ingestion (mrblip/checkpoint.py), CPU only.

* reference_keys(cfg) against the (key, shape) manifest of the REFERENCE's own BLIP2_MR.state_dict() at real width/depth (tests/golden/
  mr_c1.npz, captured by make_golden_c1.py): every reference key is consumed or listed as ignorable with a reason, nothing is missing.
* interpolate_pos_embed against the reference's function (eva_vit.py:373-394) — golden produced in the build container
  (tests/golden/make_golden_ckpt.py -> pos_embed_interp.npz).
* assemble_state_dict: eva_vit_g-style file without prefix, BLIP-2 {"model": ...} file, HF-style T5 directory (safetensors shards),
  fine-tuned peft-named checkpoint; non-strict report (missing / unexpected / ignored).
"""
import os
import sys

import numpy as np
import pytest
import torch

from util import load_golden


def _ops_available():
    return os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mr-blip_amd", "csrc", "libmrblip_hip.so"))


pytestmark = pytest.mark.skipif(not _ops_available(), reason="libmrblip_hip.so not built (run python __graft_entry__.py)")


def _c1_cfg():
    from mrblip.engine import EngineConfig
    return EngineConfig(d_model=768, d_kv=64, t5_heads=12, d_ff=2048, t5_layers=12, t5_dec_layers=12)


def test_reference_keys_cover_the_reference_state_dict():
    from mrblip import checkpoint as CK

    man = {k: tuple(s) for k, s in load_golden("mr_c1")["manifest"]}
    want = dict(CK.reference_keys(_c1_cfg()))
    assert len(want) > 1000
    for k, shape in want.items():
        assert k in man, f"engine consumes {k}, which the reference does not have"
        assert man[k] == tuple(shape), (k, man[k], shape)
    rep = CK.coverage_report(_c1_cfg(), {k: torch.empty(s, device="meta") for k, s in man.items()})
    assert rep["missing"] == [] and rep["bad_shape"] == []
    assert rep["unexpected"] == [], rep["unexpected"][:10]            # every other reference key is ignorable for a stated reason
    assert set(rep["ignored"]) == {"t5_model.encoder.embed_tokens.weight", "t5_model.decoder.embed_tokens.weight"}


def test_peft_named_state_dict_is_consumed():
    from mrblip import checkpoint as CK

    cfg = _c1_cfg()
    sd = {k: torch.empty(s, device="meta") for k, s in CK.reference_keys(cfg, peft=True)}
    assert any(".base_layer.weight" in k for k in sd) and any(".lora_A.default.weight" in k for k in sd)
    rep = CK.coverage_report(cfg, sd)
    assert rep["missing"] == [] and rep["unexpected"] == [] and rep["bad_shape"] == []


def test_interpolate_pos_embed_matches_reference():
    from mrblip import checkpoint as CK

    g = load_golden("pos_embed_interp")
    out = CK.interpolate_pos_embed(torch.from_numpy(g["pos_embed"]), int(g["num_patches"]))
    assert out.shape == g["out"].shape
    assert np.abs(out.numpy() - g["out"]).max() < 1e-6
    same = CK.interpolate_pos_embed(torch.from_numpy(g["out"]), int(g["num_patches"]))
    assert torch.equal(same, torch.from_numpy(g["out"]))              # equal grids: untouched


def test_assemble_from_reference_style_files(tmp_path):
    from safetensors.torch import save_file
    from mrblip import checkpoint as CK
    from mrblip.engine import EngineConfig

    cfg = EngineConfig.tiny()
    keys = CK.reference_keys(cfg)
    g = torch.Generator().manual_seed(0)
    full = {k: torch.randn(s, generator=g) * 0.05 for k, s in keys}
    # eva_vit_g.pth: no prefix, a LARGER position grid (6x6 -> interpolated to the model's 4x4), plus a classification head nobody reads
    vit = {k[len("visual_encoder."):]: v for k, v in full.items() if k.startswith("visual_encoder.")}
    vit["pos_embed"] = torch.randn(1, 37, cfg.vit_dim, generator=g)
    vit["head.weight"] = torch.zeros(3, cfg.vit_dim)
    torch.save(vit, tmp_path / "eva_vit_g.pth")
    blip2 = {k: v for k, v in full.items() if k.startswith(("Qformer.", "query_tokens", "ln_vision.", "t5_proj."))}
    blip2["Qformer.cls.predictions.bias"] = torch.zeros(7)
    torch.save({"model": blip2}, tmp_path / "blip2_pretrained_flant5xl.pth")
    t5dir = tmp_path / "flan-t5"
    t5dir.mkdir()
    t5 = {k[len("t5_model."):]: v.contiguous() for k, v in full.items() if k.startswith("t5_model.")}
    t5["encoder.embed_tokens.weight"] = t5["shared.weight"].clone()
    names = sorted(t5)
    save_file({k: t5[k] for k in names[: len(names) // 2]}, str(t5dir / "model-00001-of-00002.safetensors"))
    save_file({k: t5[k] for k in names[len(names) // 2:]}, str(t5dir / "model-00002-of-00002.safetensors"))
    sd, rep = CK.assemble_state_dict(cfg, vit=str(tmp_path / "eva_vit_g.pth"), blip2=str(tmp_path / "blip2_pretrained_flant5xl.pth"), t5=str(t5dir))
    assert rep["missing"] == [] and rep["bad_shape"] == [] and rep["unexpected"] == []
    assert set(rep["ignored"]) == {"visual_encoder.head.weight", "Qformer.cls.predictions.bias", "t5_model.encoder.embed_tokens.weight"}
    assert sd["visual_encoder.pos_embed"].shape == (1, 17, cfg.vit_dim)
    assert torch.equal(sd["visual_encoder.pos_embed"][:, 0], vit["pos_embed"][:, 0])           # cls position kept
    assert torch.equal(sd["t5_model.lm_head.weight"], full["t5_model.lm_head.weight"])
    # a fine-tuned (trainable-only, peft-named) checkpoint on top
    ft = {"t5_proj.weight": torch.ones(cfg.d_model, cfg.qf_dim),
          "t5_model.base_model.model.lm_head.lora_A.default.weight": torch.ones(8, cfg.d_model)}
    torch.save({"model": ft}, tmp_path / "checkpoint_best.pth")
    sd2, rep2 = CK.assemble_state_dict(cfg, vit=str(tmp_path / "eva_vit_g.pth"), blip2=str(tmp_path / "blip2_pretrained_flant5xl.pth"), t5=str(t5dir),
                                       finetuned=str(tmp_path / "checkpoint_best.pth"))
    assert rep2["missing"] == [] and rep2["unexpected"] == [] and bool((sd2["t5_proj.weight"] == 1).all())
    # missing pieces are reported, not raised (the reference's non-strict semantics); a wrong path raises the reference's message
    _, rep3 = CK.assemble_state_dict(cfg, vit=str(tmp_path / "eva_vit_g.pth"))
    assert any(k.startswith("t5_model.") for k in rep3["missing"]) and any(k.startswith("Qformer.") for k in rep3["missing"])
    with pytest.raises(RuntimeError, match="checkpoint url or path is invalid"):
        CK.load_file(str(tmp_path / "nope.pth"))
