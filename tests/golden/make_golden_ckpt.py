"""Golden for checkpoint ingestion: the reference's ``interpolate_pos_embed`` (eva_vit.py:373-394) run on a seeded position table.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ckpt.py   -> pos_embed_interp.npz"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "mr-blip_amd"))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from mrblip.tokenizer import FixtureTokenizer  # noqa: E402


def main():
    R = ref_shim.install(FixtureTokenizer, dict(vit=dict(embed_dim=32, depth=1, num_heads=2), bert={}, t5={}))
    eva = R["eva"]
    g = torch.Generator().manual_seed(7)
    pe = torch.randn(1, 1 + 16 * 16, 32, generator=g)          # a 16x16 checkpoint grid (224 px) ...
    model = types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=24 * 24), pos_embed=torch.zeros(1, 1 + 24 * 24, 32))  # ... into 24x24 (336 px)
    ck = {"pos_embed": pe.clone()}
    eva.interpolate_pos_embed(model, ck)
    np.savez_compressed(os.path.join(HERE, "pos_embed_interp.npz"), pos_embed=pe.numpy(), num_patches=np.int64(24 * 24), out=ck["pos_embed"].numpy())
    print("wrote pos_embed_interp.npz", ck["pos_embed"].shape)


if __name__ == "__main__":
    main()
