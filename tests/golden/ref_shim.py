"""Throw-away compatibility shim that makes the reference's hot-path modules importable in the BUILD
container (transformers 5.x, no timm/peft/omegaconf/wandb/av/iopath/torchvision; no network).

Used ONLY by ``make_golden.py`` to produce the fixtures under ``tests/golden/``.  Nothing here (nor
``/root/reference``) is needed or read at test/bench time: the GPU box has neither.

Recipe: SURVEY.md Appendix A.  Run with PYTHONDONTWRITEBYTECODE=1 so nothing is written under
/root/reference.
"""
import os
import sys
import types
from functools import partial

import torch
import torch.nn as nn

REF = os.environ.get("MRBLIP_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True


def _ns(name, path=None):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    sys.modules[name] = m
    return m


def install(tokenizer_factory, tiny):
    """tiny: dict with 'vit', 'bert', 't5' tiny-config kwargs used by the no-network constructors."""
    import transformers  # before the stub modules below: its import probes torchvision etc.
    import transformers.modeling_utils  # noqa
    import transformers.pytorch_utils  # noqa
    from transformers.models.t5 import configuration_t5  # noqa
    from transformers.models.bert import configuration_bert  # noqa
    import transformers.activations, transformers.modeling_outputs, transformers.file_utils  # noqa

    # 1. namespace packages (skip every __init__.py of the reference)
    for pkg in [
        "lavis", "lavis.common", "lavis.models", "lavis.models.blip2_models", "lavis.models.blip2_mr_models",
        "lavis.tasks", "lavis.processors", "lavis.datasets", "lavis.datasets.datasets", "lavis.runners",
    ]:
        _ns(pkg, os.path.join(REF, *pkg.split(".")))

    # 3. stub modules
    timm = _ns("timm", "/nonexistent")
    tm = _ns("timm.models", "/nonexistent")
    tl = _ns("timm.models.layers")
    tl.drop_path = lambda x, p=0.0, training=False: x
    tl.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)

    def trunc_normal_(t, std=1.0, **kw):
        return nn.init.trunc_normal_(t, std=std)

    tl.trunc_normal_ = trunc_normal_
    tr = _ns("timm.models.registry")
    tr.register_model = lambda f: f
    th = _ns("timm.models.hub")
    th.download_cached_file = lambda *a, **k: None
    th.get_cache_dir = lambda *a, **k: "/tmp"
    timm.models = tm
    tm.layers, tm.registry, tm.hub = tl, tr, th

    wandb = _ns("wandb")
    wandb.run = None
    wandb.log = lambda *a, **k: None
    _ns("av")
    peft = _ns("peft")
    peft.LoraConfig = lambda **k: k

    def get_peft_model(model, cfg):
        model.print_trainable_parameters = lambda: None
        return model

    peft.get_peft_model = get_peft_model
    om = _ns("omegaconf")
    om.OmegaConf = type("OmegaConf", (), {})
    _ns("iopath", "/nonexistent")
    _ns("iopath.common", "/nonexistent")
    d = _ns("iopath.common.download")
    d.download = lambda *a, **k: None
    f = _ns("iopath.common.file_io")
    f.file_lock = None
    f.g_pathmgr = None
    _ns("torchvision", "/nonexistent")
    _ns("torchvision.datasets", "/nonexistent")
    tvu = _ns("torchvision.datasets.utils")
    for n in ["check_integrity", "download_file_from_google_drive", "extract_archive"]:
        setattr(tvu, n, lambda *a, **k: None)
    bp = _ns("lavis.processors.blip_processors")

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    bp.Blip2VideoTrainProcessor = _Dummy
    bp.BlipVideoEvalProcessor = _Dummy

    # 2. transformers 4 -> 5 gaps
    import transformers
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    def find_pruneable_heads_and_indices(heads, n_heads, head_size, already_pruned_heads):
        mask = torch.ones(n_heads, head_size)
        heads = set(heads) - already_pruned_heads
        for head in heads:
            head = head - sum(1 if h < head else 0 for h in already_pruned_heads)
            mask[head] = 0
        mask = mask.view(-1).contiguous().eq(1)
        index = torch.arange(len(mask))[mask].long()
        return heads, index

    for m in (pu, mu):
        if not hasattr(m, "find_pruneable_heads_and_indices"):
            m.find_pruneable_heads_and_indices = find_pruneable_heads_and_indices
    for n in ("prune_linear_layer", "apply_chunking_to_forward"):
        if not hasattr(mu, n):
            setattr(mu, n, getattr(pu, n))
    mp = _ns("transformers.utils.model_parallel_utils")
    mp.assert_device_map = lambda *a, **k: None
    mp.get_device_map = lambda *a, **k: None

    PM = mu.PreTrainedModel
    PM.get_head_mask = lambda self, head_mask, n, is_attention_chunked=False: [None] * n

    def invert_attention_mask(self, m):
        if m.dim() == 3:
            e = m[:, None, :, :]
        else:
            e = m[:, None, None, :]
        e = e.to(dtype=torch.float32)
        return (1.0 - e) * torch.finfo(torch.float32).min

    PM.invert_attention_mask = invert_attention_mask

    def _init_all(self):
        self.apply(self._init_weights)

    PM.init_weights = _init_all
    PM.post_init = _init_all

    # 4. no-network constructors
    transformers.T5TokenizerFast.from_pretrained = classmethod(lambda cls, *a, **k: tokenizer_factory())

    import importlib

    base_model = importlib.import_module("lavis.models.base_model")
    sys.modules["lavis.models"].BaseModel = base_model.BaseModel
    t5m = importlib.import_module("lavis.models.blip2_models.modeling_t5")
    blip2 = importlib.import_module("lavis.models.blip2_models.blip2")
    eva = importlib.import_module("lavis.models.eva_vit")
    qf = importlib.import_module("lavis.models.blip2_models.Qformer")

    def t5_config(*a, **k):
        c = t5m.T5Config(**tiny["t5"])
        c.decoder_start_token_id = 0
        c.pad_token_id = 0
        c.tie_word_embeddings = False  # flan-t5 (transformers 5 ignores the constructor kwarg)
        return c

    t5m.T5Config.from_pretrained = classmethod(lambda cls, *a, **k: t5_config())

    def t5_from_pretrained(cls, name, config=None, **k):
        torch.manual_seed(0)
        return cls(config)

    t5m.T5ForConditionalGeneration.from_pretrained = classmethod(t5_from_pretrained)

    def create_eva_vit_g(img_size=224, drop_path_rate=0.0, use_checkpoint=False, precision="fp32"):
        v = tiny["vit"]
        return eva.VisionTransformer(
            img_size=img_size, patch_size=14, use_mean_pooling=False, embed_dim=v["embed_dim"], depth=v["depth"],
            num_heads=v["num_heads"], mlp_ratio=4.3637, qkv_bias=True, drop_path_rate=drop_path_rate,
            norm_layer=partial(nn.LayerNorm, eps=1e-6), use_checkpoint=use_checkpoint,
        )

    blip2.create_eva_vit_g = create_eva_vit_g

    def init_Qformer(cls, num_query_token, vision_width):
        cfg = qf.BertConfig(**tiny["bert"])
        cfg.encoder_width = vision_width
        cfg.add_cross_attention = True
        cfg.cross_attention_freq = 2
        cfg.query_length = num_query_token
        q = qf.BertLMHeadModel(cfg)
        qt = nn.Parameter(torch.zeros(1, num_query_token, cfg.hidden_size))
        qt.data.normal_(mean=0.0, std=cfg.initializer_range)
        return q, qt

    blip2.Blip2Base.init_Qformer = classmethod(init_Qformer)

    mr = importlib.import_module("lavis.models.blip2_mr_models.blip2_mr")
    mru = importlib.import_module("lavis.models.blip2_mr_models.utils")
    optims = importlib.import_module("lavis.common.optims")
    return dict(t5m=t5m, blip2=blip2, eva=eva, qf=qf, mr=mr, mru=mru, optims=optims, t5_config=t5_config)
