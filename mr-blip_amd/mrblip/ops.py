"""ctypes bindings of ``libmrblip_hip.so`` (include/mrblip_hip.h) over torch device tensors.

PyTorch is plumbing here (device memory, streams); every op below launches a hand-written gfx950 kernel on the
current torch stream.  There is NO fallback: if the library is missing the import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRBLIP_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libmrblip_hip.so")  # (override: A/B of experimental builds)


class MrblipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise MrblipError(
            f"{LIB_PATH} not found: build it with `python mr-blip_amd/csrc/build.py` (hipcc --offload-arch=gfx950). "
            "The HIP extension is the product; there is no CPU/PyTorch fallback."
        )
    lib = C.CDLL(LIB_PATH)
    lib.mrblip_last_error.restype = C.c_char_p
    return lib


_lib = _load()

vp, ll, i32, u32, f32 = C.c_void_p, C.c_longlong, C.c_int, C.c_uint32, C.c_float


def _sig(name, *argtypes):
    fn = getattr(_lib, name)
    fn.argtypes = list(argtypes)
    fn.restype = i32
    return fn


_gemm = _sig("mrblip_gemm_bf16", vp, ll, vp, ll, vp, ll, vp, ll, i32, i32, i32, vp, ll, i32, vp, ll, vp, vp, ll, i32, i32, vp, u32, f32, i32, vp)
_gemm_f16 = _sig("mrblip_gemm_f16", vp, ll, vp, ll, vp, ll, vp, ll, i32, i32, i32, vp, ll, i32, vp, ll, vp, vp, ll, i32, i32, vp, u32, f32, i32, vp)
_ln_fwd_f16 = _sig("mrblip_layernorm_fwd_f16", vp, ll, vp, vp, i32, i32, f32, vp, ll, vp, ll, vp)
_attn_fwd_rowv_f16 = _sig("mrblip_attention_fwd_rowv_f16", vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp)
_patchify_f16 = _sig("mrblip_patchify_f16", vp, vp, i32, i32, i32, i32, vp)
_patchify_u8_f16 = _sig("mrblip_patchify_u8_f16", vp, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, i32, i32, i32, i32, vp)
_ln_fwd = _sig("mrblip_layernorm_fwd", vp, ll, vp, vp, i32, i32, f32, vp, ll, vp, ll, vp)
_rms_fwd = _sig("mrblip_rmsnorm_fwd", vp, ll, vp, i32, i32, f32, vp, ll, vp, ll, vp)
_ln_bwd = _sig("mrblip_layernorm_bwd", vp, ll, vp, ll, vp, i32, i32, f32, vp, ll, vp, ll, vp, vp, vp)
_rms_bwd = _sig("mrblip_rmsnorm_bwd", vp, ll, vp, ll, vp, i32, i32, f32, vp, ll, vp, ll, vp)
_ln_bwd_cast = _sig("mrblip_layernorm_bwd_cast", vp, ll, vp, ll, vp, i32, i32, f32, vp, ll, vp, ll, vp, ll, vp, u32, f32, vp)
_rms_bwd_cast = _sig("mrblip_rmsnorm_bwd_cast", vp, ll, vp, ll, vp, i32, i32, f32, vp, ll, vp, ll, vp, ll, vp, u32, f32, vp)
_attn_fwd = _sig("mrblip_attention_fwd", vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, i32, vp, u32, f32, vp, vp)
_attn_fwd_rowv = _sig("mrblip_attention_fwd_rowv", vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp)
_attn_bwd = _sig("mrblip_attention_bwd", vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                 i32, i32, i32, i32, i32, f32, vp, vp, i32, vp, u32, f32, vp, vp)
_head_t = _sig("mrblip_head_transpose", vp, vp, vp, i32, i32, i32, i32, i32, vp, u32, f32, vp)
_colsum = _sig("mrblip_colsum", vp, ll, i32, i32, vp, vp)
_lora_tn = _sig("mrblip_lora_tn", vp, ll, vp, ll, i32, i32, i32, vp, vp, vp, vp, vp, u32, f32, vp)
_lora_pack = _sig("mrblip_lora_pack", vp, vp, vp, vp, vp, vp, i32, f32, vp)
_lora_grads = _sig("mrblip_lora_grads", vp, ll, vp, ll, vp, ll, vp, ll, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, u32, f32, vp)
_lora_grads_b_raw = _sig("mrblip_lora_grads_batched", i32, vp, vp, vp)
_lora_grads_b = lambda arr, n, sp, st: _lora_grads_b_raw(n, C.cast(arr, vp), sp, st)   # noqa: E731
_lora_down = _sig("mrblip_gemm_lora_down", vp, ll, vp, ll, i32, i32, i32, vp, ll, vp, u32, f32, vp)
_gemm_lora_dx = _sig("mrblip_gemm_lora_dx", vp, ll, vp, ll, vp, ll, vp, ll, i32, i32, i32, vp, ll, i32, vp, ll, vp, u32, f32, i32, vp)
_drop_b16 = _sig("mrblip_dropout_bf16", vp, ll, vp, ll, i32, i32, vp, u32, f32, vp)
_patchify = _sig("mrblip_patchify", vp, vp, i32, i32, i32, i32, vp)
_patchify_u8 = _sig("mrblip_patchify_u8", vp, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, i32, i32, i32, i32, vp)
_vit_asm = _sig("mrblip_vit_assemble", vp, vp, vp, vp, i32, i32, i32, vp)
_row_copy = _sig("mrblip_row_copy", vp, ll, vp, vp, ll, vp, i32, i32, i32, vp)
_mean_pool = _sig("mrblip_mean_pool", vp, vp, i32, i32, i32, vp)
_mean_pool_bwd = _sig("mrblip_mean_pool_bwd", vp, vp, i32, i32, i32, vp)
_cast_drop = _sig("mrblip_cast_dropout", vp, ll, vp, ll, vp, ll, i32, i32, vp, u32, f32, vp)
_gelu_bwd = _sig("mrblip_gelu_bwd", vp, vp, vp, ll, vp)
_gated_bwd = _sig("mrblip_gated_gelu_bwd", vp, ll, vp, ll, vp, ll, i32, i32, vp, u32, f32, vp)
_ce = _sig("mrblip_cross_entropy", vp, ll, vp, i32, i32, f32, vp, vp, ll, vp)
_ce_nv = _sig("mrblip_cross_entropy_nvalid", vp, ll, vp, i32, i32, vp, vp, vp, ll, vp)
_adamw = _sig("mrblip_adamw", vp, vp, vp, vp, ll, vp, f32, f32, f32, f32, vp)
_seed_bump = _sig("mrblip_seed_bump", vp, vp)
_prefetch = _sig("mrblip_prefetch", vp, ll, i32, vp)
_gemm_set_thin = _sig("mrblip_gemm_set_thin", vp, ll, i32, i32, u32, f32, vp, ll, u32, vp)
_gemm_stall_thin = _sig("mrblip_gemm_debug_stall_thin", i32)
_adamw_guarded = _sig("mrblip_adamw_guarded", vp, vp, vp, vp, ll, vp, f32, f32, f32, f32, vp, vp)
_gemm_set_prefetch = _sig("mrblip_gemm_set_prefetch", vp, ll, vp, ll, i32)
_lora_dx = _sig("mrblip_lora_dx_add", vp, ll, i32, vp, ll, vp, i32, i32, i32, vp, u32, f32, vp)
_cu_reserve = _sig("mrblip_gemm_set_cu_reserve", i32)
_lora_rows = _sig("mrblip_lora_rows", vp, ll, vp, ll, i32, i32, i32, vp, ll, vp, vp, u32, f32, vp)
_lora_rows_init = _sig("mrblip_lora_rows_init", vp, ll, vp, ll, i32, i32, i32, vp, ll, vp, vp, u32, f32, vp, ll, vp, ll, i32, vp)
_dec_proj = _sig("mrblip_dec_proj", vp, ll, vp, f32, vp, ll, vp, ll, vp, ll, i32, vp, ll, vp, ll, i32, i32, i32, i32, vp, ll, vp, ll, vp, ll, vp, u32, f32, u32, f32, u32, f32,
                 vp, vp, vp, i32, i32, i32, ll, ll, vp)
_dec_proj_config = _sig("mrblip_dec_proj_config", i32, i32)
_lora_dx_add_b = _sig("mrblip_lora_dx_add_batched", vp, ll, vp, ll, ll, vp, ll, i32, i32, i32, i32, vp, u32, u32, f32, vp)
_lora_rows_b = _sig("mrblip_lora_rows_batched", vp, ll, ll, vp, ll, ll, i32, i32, i32, vp, ll, ll, i32, vp, u32, u32, f32, vp)
_gemm_extra = _sig("mrblip_gemm_set_extra", vp, vp, vp, i32, i32, i32, ll, ll, ll, i32, i32)
_attn_split_ws = _sig("mrblip_attention_set_split_workspace", vp, ll, i32)
_gemm_ksplit = _sig("mrblip_gemm_ksplit", vp, ll, vp, ll, vp, ll, vp, ll, i32, i32, i32, vp, ll, ll, i32, i32, i32, vp)
_rms_bwd_parts = _sig("mrblip_rmsnorm_bwd_parts", vp, ll, i32, ll, i32, u32, f32, vp, ll, vp, i32, i32, f32, vp, ll, vp, ll, vp, ll, vp, u32, f32, vp)
_rms_bwd_parts_g = _sig("mrblip_rmsnorm_bwd_parts_g", vp, ll, i32, ll, i32, u32, f32, vp, ll, vp, i32, i32, f32, vp, ll, vp, ll, vp, ll, vp, u32, f32, vp, ll, vp, ll, vp)
_gemm_clear = _sig("mrblip_gemm_clear_one_shots")
_set_reduce_ws = _sig("mrblip_set_reduce_workspace", vp, ll)
_lib.mrblip_reduce_workspace_bytes.restype = ll
_gated_bwd_parts = _sig("mrblip_gated_gelu_bwd_parts", vp, vp, ll, vp, ll, vp, ll, i32, i32, vp, u32, f32, u32, f32, vp)
_sum_parts = _sig("mrblip_sum_parts", vp, ll, ll, i32, vp, ll, vp, ll, i32, i32, vp)
_rms_lora = _sig("mrblip_rmsnorm_lora_fwd", vp, ll, vp, i32, i32, f32, vp, ll, vp, ll, i32, vp, ll, vp, u32, f32, vp)



class QformerLayer(C.Structure):
    """mrblip_qformer_layer (include/mrblip_hip.h)"""
    _fields_ = ([(n, vp) for n in ("qkv_w", "so_w", "cq_w", "co_w", "i_w", "o_w", "qkv_b", "so_b", "s_lnw", "s_lnb", "cq_b", "co_b", "c_lnw", "c_lnb",
                                   "i_b", "o_b", "o_lnw", "o_lnb", "x_in", "x_out", "xb_out")] + [("ldxb", ll), ("qkv", vp), ("o", vp), ("ldo", ll)] +
                [(n, vp) for n in ("lse", "y", "qc", "oc", "lsec", "y2", "kv", "vt", "hpre", "y3")] +
                [("F", i32), ("Tv", i32), ("Tvp", i32), ("has_cross", i32), ("seed_ptr", vp), ("p_drop", f32)] +
                [(n, u32) for n in ("site_sattn", "site_so", "site_cattn", "site_co", "site_ffn")] + [("eps", f32)])


_qf_layer_fwd = _sig("mrblip_qformer_layer_fwd", C.POINTER(QformerLayer), vp)

EXPORTS = [
    "mrblip_last_error", "mrblip_abi_version", "mrblip_gemm_bf16", "mrblip_layernorm_fwd", "mrblip_rmsnorm_fwd",
    "mrblip_layernorm_bwd", "mrblip_rmsnorm_bwd", "mrblip_rmsnorm_bwd_cast", "mrblip_attention_fwd", "mrblip_attention_fwd_rowv", "mrblip_attention_bwd", "mrblip_head_transpose",
    "mrblip_patchify", "mrblip_vit_assemble", "mrblip_row_copy", "mrblip_mean_pool", "mrblip_mean_pool_bwd",
    "mrblip_cast_dropout", "mrblip_gelu_bwd", "mrblip_gated_gelu_bwd", "mrblip_cross_entropy", "mrblip_cross_entropy_nvalid", "mrblip_adamw", "mrblip_adamw_guarded", "mrblip_gemm_debug_stall_thin",
    "mrblip_seed_bump", "mrblip_prefetch", "mrblip_gemm_set_prefetch", "mrblip_gemm_set_thin", "mrblip_lora_dx_add", "mrblip_dropout_bf16", "mrblip_colsum", "mrblip_lora_pack", "mrblip_lora_tn",
    "mrblip_lora_grads", "mrblip_lora_grads_batched", "mrblip_gemm_lora_down", "mrblip_gemm_lora_dx", "mrblip_patchify_u8",
    "mrblip_gemm_set_cu_reserve", "mrblip_lora_rows", "mrblip_rmsnorm_lora_fwd", "mrblip_lora_rows_init", "mrblip_dec_proj", "mrblip_dec_proj_config", "mrblip_lora_dx_add_batched", "mrblip_lora_rows_batched", "mrblip_gemm_set_extra", "mrblip_attention_set_split_workspace",
    "mrblip_gemm_ksplit", "mrblip_rmsnorm_bwd_parts", "mrblip_rmsnorm_bwd_parts_g", "mrblip_set_reduce_workspace", "mrblip_reduce_workspace_bytes", "mrblip_gemm_clear_one_shots", "mrblip_gated_gelu_bwd_parts", "mrblip_sum_parts",
    "mrblip_qformer_layer_fwd", "mrblip_layernorm_bwd_cast",
    "mrblip_gemm_f16", "mrblip_layernorm_fwd_f16", "mrblip_attention_fwd_rowv_f16", "mrblip_patchify_f16", "mrblip_patchify_u8_f16",
]


launch_count = 0  # C-ABI calls so far (each is one kernel launch, mrblip_attention_bwd two): bench.py reports the per-step difference


def _chk(rc: int):
    global launch_count
    launch_count += 1
    if rc != 0:
        raise MrblipError(_lib.mrblip_last_error().decode())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) if os.environ.get("MRB_SLOW_STREAM_LOOKUP") is None else None

def _stream() -> int:
    """raw hipStream_t of torch's current stream.  ~2700 ops per step ask for it: torch's C accessor (no Stream object per call) takes
    ~0.2 us against ~2.5 us for torch.cuda.current_stream().cuda_stream — 6 ms of host time per step"""
    if _raw_stream is None:
        return torch.cuda.current_stream().cuda_stream
    return _raw_stream(torch.cuda.current_device())


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _ld(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.stride(0)


def _req(t, dtype, name):
    if t.dtype != dtype or not t.is_cuda or t.stride(-1) != 1:
        raise MrblipError(f"{name}: expected a cuda {dtype} tensor contiguous in its last dim, got {t.dtype} {tuple(t.shape)} {t.stride()}")


class Dropout:
    """(seed tensor on device, call-site id, p).  ``None`` disables dropout."""

    __slots__ = ("seed", "site", "p")

    def __init__(self, seed: torch.Tensor, site: int, p: float):
        self.seed, self.site, self.p = seed, site & 0xFFFFFFFF, float(p)


def _d(d: Optional[Dropout]):
    if d is None or d.p <= 0.0:
        return None, 0, 0.0
    return d.seed.data_ptr(), d.site, d.p


def qformer_layer_fwd(fields: dict, *, seed: Optional[torch.Tensor], p_drop: float, sites: Sequence[int], eps: float):
    """One Q-Former layer's query branch in one launch (mrblip_qformer_layer_fwd).  ``fields``: tensor per pointer field of
    mrblip_qformer_layer (None where the layer has no cross-attention) plus the ints F, Tv, Tvp, has_cross, ldxb, ldo."""
    a = QformerLayer()
    for name, typ in QformerLayer._fields_:
        v = fields.get(name)
        if typ is vp and name != "seed_ptr":
            setattr(a, name, None if v is None else v.data_ptr())
        elif name in ("F", "Tv", "Tvp", "has_cross", "ldxb", "ldo"):
            setattr(a, name, int(v or 0))
    on = seed is not None and p_drop > 0.0
    a.seed_ptr = seed.data_ptr() if on else None
    a.p_drop = float(p_drop) if on else 0.0
    a.site_sattn, a.site_so, a.site_cattn, a.site_co, a.site_ffn = (int(x) & 0xFFFFFFFF for x in sites)
    a.eps = float(eps)
    _chk(_qf_layer_fwd(C.byref(a), _stream()))


# ------------------------------------------------------------------------------------------------ GEMM
def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, aext=None, wext=None, out2=None, bias=None, residual=None,
         act: int = 0, gated: bool = False, drop: Optional[Dropout] = None, tile_cfg: int = 0, K: Optional[int] = None,
         cu_reserve: Optional[int] = None, k_splits: int = 0, tout=None, t_rows: int = 0, ext_group_n: int = 0, thin=None):
    """out[M,N] = a[M,K] @ w[N,K]^T (+ aext @ wext^T) with the fused epilogue of mrblip_gemm_bf16.  cu_reserve: CUs a persistent
    tile kernel leaves to other streams (None = the calling thread's ``gemm_cu_reserve`` context, default 0).  k_splits > 1 (M <= 32,
    fp32 out, no residual): the skinny kernel's blocks split K and ADD into ``out``, which the caller pre-initialised.
    tout / t_rows (bf16 out, plain or bias epilogue): up to three [B, H, 64, Spad] tensors receiving the head-transposed copies of the
    output's consecutive column ranges of width H * 64, rows being b * t_rows + s (see ``dec_proj``).  ext_group_n: output-column group
    g = n // ext_group_n takes columns [64 g, 64 g + 64) of ``aext`` as its K extension (one GEMM for several LoRA groups).
    thin = (acat [R, K'], K', Dropout or None): the launch computes aext[:, :R] = dropout(a)[:, :K'] @ acat^T itself (``lora_rows`` without a
    launch of its own: the first workgroups of the GEMM do it while the tiles run; M > 64)."""
    try:     # (the one-shots set below — and a prefetch range the caller set before this call — belong to THIS launch: an exception on the way drops them)
        f16 = a.dtype == torch.float16   # IEEE fp16 operands (the fp16-operand ViT): both operands, and a 16-bit output, are fp16
        _req(a, torch.float16 if f16 else torch.bfloat16, "gemm.a"); _req(w, a.dtype, "gemm.w")
        if out.dtype != torch.float32 and out.dtype != a.dtype:
            raise MrblipError(f"gemm: 16-bit output must have the operands' dtype ({a.dtype}), got {out.dtype}")
        M = a.shape[0]
        N = w.shape[0]
        K = a.shape[1] if K is None else K
        sp, site, p = _d(drop)
        reserve = getattr(_tls, "cu_reserve", 0) if cu_reserve is None else cu_reserve
        _set_gemm_extra(tout, t_rows, ext_group_n)
        if thin is not None:
            acat, tk, tdrop = thin
            tsp, tsite, tp = _d(tdrop)
            if sp is None:
                sp = tsp          # the launch's one seed pointer (the epilogue's own dropout stays off: p = 0)
            flags, epoch = _thin_flags(a.device, (M + 15) // 16)
            if _gemm_set_thin(_p(acat), _ld(acat), acat.shape[0], int(tk), tsite, tp, _p(flags), flags.numel(), epoch, _p(thin_error_word(a.device))) != 0:
                raise MrblipError(_lib.mrblip_last_error().decode())
        _chk((_gemm_f16 if f16 else _gemm)(_p(a), _ld(a), _p(w), _ld(w), _p(aext), _ld(aext), _p(wext), _ld(wext), M, N, K, _p(out), _ld(out),
                   1 if out.dtype == torch.float32 else 0, _p(out2), _ld(out2), _p(bias), _p(residual), _ld(residual), act,
                   1 if gated else 0, sp, site, p, (tile_cfg & 0xff) | ((int(reserve) & 0x1ff) << 8) | ((int(k_splits) & 0xf) << 17), _stream()))
    except BaseException:
        _clear_one_shots()
        raise
    return out


_tls = threading.local()
_thin_state = {}
_reduce_state = {}


def _reduce_ws(device):
    """Register the ordered-reduction workspace of the CURRENT stream with the library (mrblip_set_reduce_workspace) for the launch that
    follows: cross_entropy, colsum and layernorm_bwd(dgamma) add their partial sums in a fixed order through caller-owned scratch + tickets,
    and launches on different streams must not share them — one zeroed buffer per (device, stream), allocated once and never moved (a
    captured graph keeps its address)."""
    key = (device.index, _stream())
    ws = _reduce_state.get(key)
    if ws is None:
        ws = _reduce_state[key] = torch.zeros(int(_lib.mrblip_reduce_workspace_bytes()), dtype=torch.uint8, device=device)
    if _set_reduce_ws(ws.data_ptr(), ws.numel()) != 0:
        raise MrblipError(_lib.mrblip_last_error().decode())


THIN_FLAG_WORDS = 1 << 17     # flag words per (device, stream): one per 16 rows -> GEMMs of up to ~2 M rows


def _thin_flags(device, n: int):
    """(flag words, fresh epoch) for a GEMM with the thin role on the CURRENT stream: launches of one stream are ordered, so they share a
    buffer and tell their flags apart by the epoch; another stream gets its own buffer.  The buffer is allocated ONCE at its maximum size
    and never replaced: a captured graph bakes its address (and a fill of it) in, and a replaced buffer would be freed under it."""
    key = (device.index, _stream())
    st = _thin_state.get(key)
    if st is None:      # flags | ticket, finished count (the kernel keeps them at zero between launches) | spare
        st = _thin_state[key] = [torch.zeros(THIN_FLAG_WORDS, dtype=torch.int32, device=device), 0]
    if n + 4 > st[0].numel():
        raise MrblipError(f"gemm thin role: {n} row blocks exceed the flag buffer ({st[0].numel() - 4}); run this launch without the thin role")
    st[1] = st[1] % 0x7fffffff + 1
    return st[0], st[1]


_thin_err = {}


def thin_error_word(device) -> torch.Tensor:
    """THE error word of the in-GEMM thin role on ``device`` (one int32, shared by every stream's launches): a consumer tile whose bounded
    wait for its producer workgroups ran out stores 0xffffffff here (csrc/gemm.hip).  It is also the ``guard`` of the fused AdamW
    (``adamw(..., guard=)``: a non-zero word makes the optimizer step a no-op ON THE DEVICE), and MrBlipEngine.check_thin_role() reads it
    one step late and raises — a protocol failure skips the update and stops the run instead of training on a corrupted step."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    t = _thin_err.get(idx)
    if t is None:
        t = _thin_err[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
    return t


def thin_flags_reset(device):
    """clear the thin role's flag words of the CURRENT stream (creating them if needed).  Called at the head of a stream capture: the
    launches' epochs are kernel arguments, so a replayed graph would find the previous replay's flags already set — the captured fill
    clears them on every replay (include/mrblip_hip.h: mrblip_gemm_set_thin)."""
    key = (device.index, _stream())
    st = _thin_state.get(key)
    if st is None:
        _thin_state[key] = [torch.zeros(THIN_FLAG_WORDS, dtype=torch.int32, device=device), 0]
    else:
        st[0].zero_()


def gemm_thin_timeouts() -> int:
    """devices whose thin-role error word is set: a tile's bounded wait for the thin role ran out (synchronises; 0 in every correct run)"""
    return sum(int(t.item() != 0) for t in _thin_err.values())


def gemm_thin_clear():
    """reset the error words (tests of the failure path)"""
    for t in _thin_err.values():
        t.zero_()


class gemm_debug_stall_thin:
    """TEST HOOK: ``with gemm_debug_stall_thin():`` the thin-role workgroups of this thread's GEMM launches exit without publishing, so the
    consumer tiles run into their bounded wait (about 0.2 s per launch) and set the error word"""

    def __enter__(self):
        self.prev = _gemm_stall_thin(1)
        return self

    def __exit__(self, *exc):
        _gemm_stall_thin(self.prev)
        return False


def _clear_one_shots():
    """drop the calling thread's pending one-shot GEMM extras (head-transposed copies, grouped K extension, prefetch range) after a
    failure between their setters and the launch they were meant for"""
    _gemm_clear()


def _set_gemm_extra(tout, t_rows: int, ext_group_n: int = 0):
    """one-shot extras of this thread's next GEMM launch (mrblip_gemm_set_extra): head-transposed copies and / or the grouped K extension"""
    if (tout is None or (not isinstance(tout, torch.Tensor) and all(x is None for x in tout))) and not ext_group_n:
        return
    t = [None, None, None]
    t_inner = t_spad = 0
    t_bs = t_hs = 0
    t_stride = t_count = 0
    if isinstance(tout, torch.Tensor):                     # ONE buffer [ranges, B, H, 64, Spad]: range j -> tout[j]
        assert tout.dim() == 5 and tout.shape[3] == 64 and tout.is_contiguous()
        t[0] = tout
        t_count, t_stride = tout.shape[0], tout.stride(0)
        t_inner, t_spad, t_bs, t_hs = tout.shape[2] * 64, tout.shape[4], tout.stride(1), tout.stride(2)
    elif tout is not None and any(x is not None for x in tout):
        ref = next(x for x in tout if x is not None)      # [B, H, 64, Spad]
        assert ref.shape[2] == 64 and all(x is None or (x.shape == ref.shape and x.is_contiguous()) for x in tout)
        t[:len(tout)] = list(tout)
        t_inner, t_spad, t_bs, t_hs = ref.shape[1] * 64, ref.shape[3], ref.stride(0), ref.stride(1)
    rc = _gemm_extra(_p(t[0]), _p(t[1]), _p(t[2]), t_inner, int(t_rows), t_spad, t_bs, t_hs, t_stride, t_count, int(ext_group_n))
    if rc != 0:
        raise MrblipError(_lib.mrblip_last_error().decode())


class gemm_cu_reserve:
    """``with gemm_cu_reserve(n):`` the persistent GEMM kernels this THREAD launches inside leave n CUs (a multiple of 8) to other
    streams.  The value travels with each call (bits 8..16 of the C entry's tile_cfg): the library keeps no process-global state."""

    def __init__(self, n: int):
        self.n = int(n)

    def __enter__(self):
        self.prev = getattr(_tls, "cu_reserve", 0)
        _tls.cu_reserve = self.n
        return self

    def __exit__(self, *exc):
        _tls.cu_reserve = self.prev
        return False


# ------------------------------------------------------------------------------------------------ norms
def layernorm_fwd(x, gamma, beta, eps, out_bf16=None, out_f32=None):
    """out_bf16: the 16-bit output; a torch.float16 tensor gets IEEE fp16 values (fp16-operand ViT), a bfloat16 one bf16"""
    _req(x, torch.float32, "layernorm.x")
    M, D = x.shape
    fn = _ln_fwd_f16 if (out_bf16 is not None and out_bf16.dtype == torch.float16) else _ln_fwd
    _chk(fn(_p(x), _ld(x), _p(gamma), _p(beta), M, D, eps, _p(out_bf16), _ld(out_bf16), _p(out_f32), _ld(out_f32), _stream()))


def rmsnorm_fwd(x, weight, eps, out_bf16=None, out_f32=None):
    _req(x, torch.float32, "rmsnorm.x")
    M, D = x.shape
    _chk(_rms_fwd(_p(x), _ld(x), _p(weight), M, D, eps, _p(out_bf16), _ld(out_bf16), _p(out_f32), _ld(out_f32), _stream()))


def layernorm_bwd(dy, x, gamma, eps, dx, dx_add=None, dgamma=None, dbeta=None, out_bf16=None, out_drop: Optional["Dropout"] = None):
    """out_bf16 (optional, dx-only form): also write bf16(dropout-backward(dx)) for the mask of ``out_drop`` — the next GEMM's operand — in the
    same launch (mrblip_layernorm_bwd_cast; the same bits as a cast_dropout launch behind it)"""
    M, D = x.shape
    if out_bf16 is not None:
        assert dgamma is None and dbeta is None, "layernorm_bwd: the fused cast is the frozen-weights (dx only) form"
        sp, site, p = _d(out_drop)
        _chk(_ln_bwd_cast(_p(dy), _ld(dy), _p(x), _ld(x), _p(gamma), M, D, eps, _p(dx_add), _ld(dx_add), _p(dx), _ld(dx), _p(out_bf16), _ld(out_bf16),
                          sp, site, p, _stream()))
        return
    if dgamma is not None:
        _reduce_ws(x.device)
    _chk(_ln_bwd(_p(dy), _ld(dy), _p(x), _ld(x), _p(gamma), M, D, eps, _p(dx_add), _ld(dx_add), _p(dx), _ld(dx), _p(dgamma), _p(dbeta), _stream()))


def rmsnorm_bwd(dy, x, weight, eps, dx, dx_add=None, out_bf16=None, out_drop: Optional["Dropout"] = None, ext_drop: Optional["Dropout"] = None, ext_part: bool = False,
                g_prod=None):
    """out_bf16 (optional): also write bf16(dropout-backward(dx)) for the mask of ``out_drop`` — the next GEMM's operand, saving the
    separate cast_dropout launch and its read of dx.  dy of 3 dims [parts, M, D]: partial products (gemm_ksplit), added in part order;
    ext_part: the last one is the LoRA term, added under the keep mask of ``ext_drop``.
    g_prod = (g_b bf16 [8, >= D], g_out bf16 [M, >= 8]) (needs out_bf16): also g_out[:, :8] = out_bf16 @ g_b[:, :D]^T — the LoRA "g" product
    of the projection that consumes out_bf16 (what ``lora_rows(out_bf16, g_b, g_out, D)`` would compute in a launch of its own)."""
    M, D = x.shape
    if g_prod is not None:
        g_b, g_out = g_prod
        assert out_bf16 is not None and g_b.shape[0] == 8 and g_b.dtype == torch.bfloat16 and g_out.dtype == torch.bfloat16
        parts = dy if dy.dim() == 3 else dy.unsqueeze(0)
        sp, site, p = _d(out_drop)
        esp, esite, ep = _d(ext_drop)
        assert parts.stride(2) == 1 and (sp == esp or not sp or not esp)
        _chk(_rms_bwd_parts_g(_p(parts), parts.stride(1), parts.shape[0], parts.stride(0), 1 if ext_part else 0, esite, ep, _p(x), _ld(x), _p(weight), M, D, eps,
                              _p(dx_add), _ld(dx_add), _p(dx), _ld(dx), _p(out_bf16), _ld(out_bf16), sp or esp, site, p,
                              _p(g_b), _ld(g_b), _p(g_out), _ld(g_out), _stream()))
        return
    if dy.dim() == 3:
        sp, site, p = _d(out_drop)
        esp, esite, ep = _d(ext_drop)
        assert dy.stride(2) == 1 and (sp == esp or not sp or not esp)
        _chk(_rms_bwd_parts(_p(dy), dy.stride(1), dy.shape[0], dy.stride(0), 1 if ext_part else 0, esite, ep, _p(x), _ld(x), _p(weight), M, D, eps,
                            _p(dx_add), _ld(dx_add), _p(dx), _ld(dx), _p(out_bf16), _ld(out_bf16), sp or esp, site, p, _stream()))
        return
    if out_bf16 is None:
        _chk(_rms_bwd(_p(dy), _ld(dy), _p(x), _ld(x), _p(weight), M, D, eps, _p(dx_add), _ld(dx_add), _p(dx), _ld(dx), _stream()))
    else:
        sp, site, p = _d(out_drop)
        _chk(_rms_bwd_cast(_p(dy), _ld(dy), _p(x), _ld(x), _p(weight), M, D, eps, _p(dx_add), _ld(dx_add), _p(dx), _ld(dx), _p(out_bf16), _ld(out_bf16),
                           sp, site, p, _stream()))


# ------------------------------------------------------------------------------------------------ attention
def _strides3(t: torch.Tensor):
    """t viewed as [B, S, H, D] (any strides, D contiguous) -> ctypes {batch, head, row} strides."""
    assert t.dim() == 4 and t.stride(3) == 1
    return (ll * 3)(t.stride(0), t.stride(2), t.stride(1))


def rup32(n: int) -> int:
    return (n + 31) // 32 * 32


def head_transpose(x: torch.Tensor, out: Optional[torch.Tensor] = None, spad: int = 0, drop: Optional[Dropout] = None) -> torch.Tensor:
    """x: [B,S,H,D] view (bf16) -> [B,H,rup32(D),spad or rup32(S)] zero-padded transposed copy."""
    _req(x, torch.bfloat16, "head_transpose.x")
    B, S, H, D = x.shape
    if out is None:
        out = torch.empty(B, H, rup32(D), spad or rup32(S), dtype=torch.bfloat16, device=x.device)
    sp, site, p = _d(drop)
    _chk(_head_t(_p(x), _strides3(x), _p(out), B, H, S, D, spad, sp, site, p, _stream()))
    return out


def colsum(x, out):
    M, N = x.shape
    _reduce_ws(x.device)
    _chk(_colsum(_p(x), _ld(x), M, N, _p(out), _stream()))


def lora_tn(Y, U, outs, col0, ncols, lds, drop: Optional[Dropout] = None):
    """outs[j][r, c - col0[j]] += sum_m U[m, 8j + r] * drop(Y)[m, c] for col0[j] <= c < col0[j] + ncols[j]; outs: fp32 [8, *] views."""
    M, Cc = Y.shape
    n = len(outs)
    sp, site, p = _d(drop)
    _chk(_lora_tn(_p(Y), _ld(Y), _p(U), _ld(U), M, Cc, 8 * n, (vp * n)(*[o.data_ptr() for o in outs]), (i32 * n)(*col0), (i32 * n)(*ncols),
                  (ll * n)(*lds), sp, site, p, _stream()))


def lora_pack(flat, acat, wext, bblk, acatt, desc, n_adapters, scale=1.0):
    _chk(_lora_pack(_p(flat), _p(acat), _p(wext), _p(bblk), _p(acatt), _p(desc), n_adapters, scale, _stream()))


def lora_grads(dy, u, x, g, dBt, b_col0, b_ncols, dA, K, drop: Optional[Dropout] = None):
    """both weight gradients of a fused LoRA group in one launch: dBt[j] += u_j^T dy[:, cols_j], dA[j] += g_j^T dropout(x[:, :K])"""
    M, N = dy.shape
    n = len(dBt)
    sp, site, p = _d(drop)
    _chk(_lora_grads(_p(dy), _ld(dy), _p(u), _ld(u), _p(x), _ld(x), _p(g), _ld(g), M, N, K, 8 * n,
                     (vp * n)(*[o.data_ptr() for o in dBt]), (i32 * n)(*b_col0), (i32 * n)(*b_ncols), (ll * n)(*b_ncols),
                     (vp * n)(*[o.data_ptr() for o in dA]), (ll * n)(*([K] * n)), sp, site, p, _stream()))


class LoraGradsJob(C.Structure):
    """mirror of MrblipLoraGradsJob (include/mrblip_hip.h)"""
    _fields_ = [("dY", vp), ("lddy", ll), ("U", vp), ("ldu", ll), ("X", vp), ("ldx", ll), ("G", vp), ("ldg", ll),
                ("M", i32), ("N", i32), ("K", i32), ("R", i32),
                ("dBt", vp * 4), ("b_col0", i32 * 4), ("b_ncols", i32 * 4), ("b_lds", ll * 4),
                ("dA", vp * 4), ("a_lds", ll * 4), ("site", u32), ("p_drop", f32)]


def lora_grads_job(dy, u, x, g, dBt, b_col0, b_ncols, dA, K, drop: Optional[Dropout] = None):
    """the arguments of one ``lora_grads`` call as a job of ``lora_grads_batched`` (a tuple: (seed tensor or None, filled struct))"""
    j = LoraGradsJob()
    j.dY, j.lddy, j.U, j.ldu, j.X, j.ldx, j.G, j.ldg = dy.data_ptr(), _ld(dy), u.data_ptr(), _ld(u), x.data_ptr(), _ld(x), g.data_ptr(), _ld(g)
    j.M, j.N, j.K, j.R = dy.shape[0], dy.shape[1], K, 8 * len(dBt)
    for i in range(len(dBt)):
        j.dBt[i], j.b_col0[i], j.b_ncols[i], j.b_lds[i] = dBt[i].data_ptr(), b_col0[i], b_ncols[i], b_ncols[i]
        j.dA[i], j.a_lds[i] = dA[i].data_ptr(), K
    seed = None
    if drop is not None and drop.p > 0.0:
        j.site, j.p_drop, seed = drop.site, drop.p, drop.seed
    return seed, j


def lora_grads_batched(jobs):
    """both weight gradients of up to 8 fused LoRA groups in ONE launch (same bits as one ``lora_grads`` call per group)"""
    n = len(jobs)
    assert 1 <= n <= 8
    seeds = {s.data_ptr(): s for s, _ in jobs if s is not None}
    assert len(seeds) <= 1, "one device seed per launch"
    arr = (LoraGradsJob * n)(*[j for _, j in jobs])
    sp = next(iter(seeds)) if seeds else None
    _chk(_lora_grads_b(arr, n, sp, _stream()))


def lora_down(x, acat, u, K, drop: Optional[Dropout] = None):
    """u[:, :R] = dropout(x[:, :K]) @ acat^T  (acat: bf16 [R, K]); dropout fused into the operand load"""
    M = x.shape[0]
    sp, site, p = _d(drop)
    _chk(_lora_down(_p(x), _ld(x), _p(acat), _ld(acat), M, acat.shape[0], K, _p(u), _ld(u), sp, site, p, _stream()))


def lora_rows(x, a, u, K, drop: Optional[Dropout] = None, seg: Optional[Sequence[int]] = None, init_dst=None, init_src=None):
    """u[:, :R] = dropout(x[:, :K]) @ a^T  (a: bf16 [R, >= K], R <= 32) — the row kernel (csrc/lora.hip).  seg: [k0, k1) per 8-row
    group of ``a`` outside of which the group is zero (the block-diagonal s*B^T of a fused LoRA group): skipped work, same result."""
    sp, site, p = _d(drop)
    R = a.shape[0]
    segp = None
    if seg is not None:
        assert len(seg) == 2 * (R // 8)
        segp = (i32 * len(seg))(*[int(v) for v in seg])
    if init_dst is None:
        _chk(_lora_rows(_p(x), _ld(x), _p(a), _ld(a), x.shape[0], R, K, _p(u), _ld(u), segp, sp, site, p, _stream()))
    else:  # + init_dst[m, :] = init_src[m, :] (or 0): the fp32 output a following K-split gemm(..., k_splits=n) accumulates into
        _req(init_dst, torch.float32, "lora_rows.init_dst")
        _chk(_lora_rows_init(_p(x), _ld(x), _p(a), _ld(a), x.shape[0], R, K, _p(u), _ld(u), segp, sp, site, p, _p(init_dst), _ld(init_dst),
                             _p(init_src), _ld(init_src), init_dst.shape[1], _stream()))


def rmsnorm_lora_fwd(x, weight, eps, out_bf16, a, u, drop: Optional[Dropout] = None):
    """T5 RMSNorm -> bf16 rows, and u[:, :R] = dropout(rows) @ a^T in the same launch"""
    _req(x, torch.float32, "rmsnorm_lora.x")
    M, D = x.shape
    sp, site, p = _d(drop)
    _chk(_rms_lora(_p(x), _ld(x), _p(weight), M, D, eps, _p(out_bf16), _ld(out_bf16), _p(a), _ld(a), a.shape[0], _p(u), _ld(u), sp, site, p, _stream()))


def dec_proj(xin, w, a, bt, u, out, K, *, x32=None, gamma=None, eps=0.0, N=None, residual=None, out2=None, gated=False,
             in_drop: Optional[Dropout] = None, out_drop: Optional[Dropout] = None, ext_drop: Optional[Dropout] = None, tout=None, t_rows: int = 0):
    """One launch for an adapted projection of <= 16 decoder rows (csrc/decproj.hip): u = dropout_in(xin) a^T (saved), out = xin w^T +
    [mask_ext (.)] u bt^T with the epilogue the output asks for (bf16 | fp32 residual + dropout_out | gated with out2 = pre-activations).
    x32 / gamma / eps: the input is RMSNorm(x32) * gamma, saved to ``xin``.  N: output columns (default w.shape[0], gated: half of it).
    tout (bf16 output only): up to three [B, H, 64, Spad] tensors (None to skip one) receiving the head-transposed copies of the output's
    consecutive column ranges of width H * 64 (what head_transpose would write), rows being b * t_rows + s."""
    R = xin.shape[0] if x32 is None else x32.shape[0]
    if N is None:
        N = w.shape[0] // 2 if gated else w.shape[0]
    mode = 2 if gated else (1 if out.dtype == torch.float32 else 0)
    seeds = [d for d in (in_drop, out_drop, ext_drop) if d is not None and d.p > 0.0]
    sp = seeds[0].seed.data_ptr() if seeds else None
    s_in, p_in = (in_drop.site, in_drop.p) if in_drop is not None else (0, 0.0)
    s_out, p_out = (out_drop.site, out_drop.p) if out_drop is not None else (0, 0.0)
    s_ext, p_ext = (ext_drop.site, ext_drop.p) if ext_drop is not None else (0, 0.0)
    t = [None, None, None]
    t_inner = t_spad = 0
    t_bs = t_hs = 0
    if tout is not None and any(x is not None for x in tout):
        ref = next(x for x in tout if x is not None)      # [B, H, 64, Spad]
        assert ref.shape[2] == 64 and all(x is None or (x.shape == ref.shape and x.is_contiguous()) for x in tout)
        t[:len(tout)] = list(tout)
        t_inner, t_spad, t_bs, t_hs = ref.shape[1] * 64, ref.shape[3], ref.stride(0), ref.stride(1)
    _chk(_dec_proj(_p(x32), _ld(x32), _p(gamma), eps, _p(xin), _ld(xin), _p(w), _ld(w), _p(a), _ld(a), a.shape[0], _p(bt), _ld(bt), _p(u), _ld(u),
                   R, N, K, mode, _p(out), _ld(out), _p(residual), _ld(residual), _p(out2), _ld(out2), sp, s_in, p_in, s_out, p_out, s_ext, p_ext,
                   _p(t[0]), _p(t[1]), _p(t[2]), t_inner, int(t_rows), t_spad, t_bs, t_hs, _stream()))


def attention_split_workspace(ws: Optional[torch.Tensor], n_split: int = 0):
    """Register (or, with None, remove) the workspace of the cross-block key split of the few-query attention form for this thread's later
    attention_fwd / attention_bwd launches (mrblip_attention_set_split_workspace): a ZEROED, 16-B aligned device tensor of at least
    16 KB + B * H * n_split * 9216 bytes; n_split = 0 lets the library choose about one block per CU."""
    rc = _attn_split_ws(None, 0, 0) if ws is None else _attn_split_ws(_p(ws), ws.numel() * ws.element_size(), int(n_split))
    if rc != 0:
        raise MrblipError(_lib.mrblip_last_error().decode())


def lora_rows_batched(x, a, u, K, groups: int, *, x_gstride: int = 0, a_gstride: int, u_gstride: int, R: int, drop: Optional[Dropout] = None, site_stride: int = 0):
    """``lora_rows`` for ``groups`` problems in one launch: group g reads x[:, g * x_gstride:] (0: the same rows), the R rows of ``a`` at
    element offset g * a_gstride, writes u[:, g * u_gstride:] and masks with call-site id drop.site + g * site_stride."""
    _req(x, torch.bfloat16, "lora_rows_batched.x")
    sp, site, p = _d(drop)
    _chk(_lora_rows_b(_p(x), _ld(x), int(x_gstride), _p(a), K if a.dim() == 1 else _ld(a), int(a_gstride), x.shape[0], int(R), int(K), _p(u), _ld(u), int(u_gstride),
                      int(groups), sp, site, int(site_stride), p, _stream()))


def lora_dx_add_batched(dx, g, a, K, groups: int, *, g_gstride: int, a_gstride: int, R: int, drop: Optional[Dropout] = None, site_stride: int = 0):
    """dx (fp32 [M, >= K]) += sum over groups of mask_g (.) (g[:, j * g_gstride : + R] @ a_j[R, K]); a_j at element offset j * a_gstride of ``a``"""
    _req(dx, torch.float32, "lora_dx_add_batched.dx")
    sp, site, p = _d(drop)
    _chk(_lora_dx_add_b(_p(dx), _ld(dx), _p(g), _ld(g), int(g_gstride), _p(a), int(a_gstride), int(R), dx.shape[0], int(K), int(groups), sp, site,
                        int(site_stride), p, _stream()))


def dec_proj_config(n_blocks: int = -1, version: int = -1) -> int:
    """Launch shape of ``dec_proj`` for <= 16 rows (mrblip_dec_proj_config): n_blocks > 0 = blocks of the streaming kernel (each owns a contiguous
    range of 16-column tiles; 0 = one per CU; < 0 = unchanged), version 0 = the one-tile-per-block kernel of round 3, 1 = streaming.  Returns the
    previous n_blocks.  Thread-local on the C side, like ``gemm_cu_reserve``."""
    return int(_dec_proj_config(int(n_blocks), int(version)))


class dec_proj_grid:
    """``with dec_proj_grid(n):`` dec_proj launches of this thread use n blocks (0: one per CU) inside"""

    def __init__(self, n: int):
        self.n = int(n)

    def __enter__(self):
        self.prev = dec_proj_config(self.n)
        return self

    def __exit__(self, *exc):
        dec_proj_config(self.prev)
        return False


def lora_dx(dy, wt, g, acatt, dx, K, residual=None, drop: Optional[Dropout] = None, tile_cfg=0, k_splits: int = 0, tout=None, t_rows: int = 0):
    """dx = dy[:, :K] @ wt^T (+ residual) + mask(drop) * (g @ acatt^T);  wt: bf16 [N_in, >=K], acatt: bf16 [N_in, 64].
    tout / t_rows (bf16 dx, no residual): head-transposed copies of dx, as in ``gemm``."""
    M = dy.shape[0]
    N = wt.shape[0]
    sp, site, p = _d(drop)
    try:
        _req(dy, torch.bfloat16, "lora_dx.dy"); _req(wt, torch.bfloat16, "lora_dx.wt")
        _set_gemm_extra(tout, t_rows)
        _chk(_gemm_lora_dx(_p(dy), _ld(dy), _p(wt), _ld(wt), _p(g), _ld(g), _p(acatt), _ld(acatt), M, N, K, _p(dx), _ld(dx),
                      1 if dx.dtype == torch.float32 else 0, _p(residual), _ld(residual) if residual is not None else 0, sp, site, p,
                      (tile_cfg & 0xff) | ((int(k_splits) & 0xf) << 17), _stream()))
    except BaseException:
        _clear_one_shots()
        raise


def dropout_bf16(x, out, drop: Optional[Dropout] = None):
    M, N = x.shape
    sp, site, p = _d(drop)
    _chk(_drop_b16(_p(x), _ld(x), _p(out), _ld(out), M, N, sp, site, p, _stream()))


def drop_bits_shape(B, H, Sq, Sk):
    """shape of the uint32 (int32 tensor) scratch carrying the attention-dropout keep mask from forward to backward"""
    return (B * H, rup32(Sk) // 32, rup32(Sq))


def attention_fwd(q, k, vt, o, lse=None, *, scale=1.0, bias_lut=None, kmask=None, causal=False, drop: Optional[Dropout] = None,
                  drop_bits=None):
    """q,o: [B,Sq,H,D] views; k: [B,Sk,H,D] view; vt: head_transpose(v); lse: [B,H,rup32(Sq)] fp32."""
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    sp, site, p = _d(drop)
    _chk(_attn_fwd(_p(q), _strides3(q), _p(k), _strides3(k), _p(vt), _p(o), _strides3(o), _p(lse), B, H, Sq, Sk, D, scale,
                   _p(bias_lut), _p(kmask), 1 if causal else 0, sp, site, p, _p(drop_bits), _stream()))


def attention_fwd_rowv(q, k, v, o, lse=None, *, scale=1.0):
    """the plain ViT forward with v as a row-major [B,S,H,D] view (a column slice of the fused qkv buffer): no head_transpose of V"""
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    if not (q.dtype == k.dtype == v.dtype == o.dtype):
        raise MrblipError("attention_fwd_rowv: q, k, v, o must share one 16-bit dtype")
    fn = _attn_fwd_rowv_f16 if q.dtype == torch.float16 else _attn_fwd_rowv
    _chk(fn(_p(q), _strides3(q), _p(k), _strides3(k), _p(v), _strides3(v), _p(o), _strides3(o), _p(lse), B, H, Sq, Sk, D, scale,
                        _stream()))


def attention_bwd(q, k, v, o, do, kt, qt, dot, lse, delta, dq, dk, dv, *, scale=1.0, bias_lut=None, kmask=None, causal=False,
                  drop: Optional[Dropout] = None, drop_bits=None):
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    sp, site, p = _d(drop)
    _chk(_attn_bwd(_p(q), _strides3(q), _p(k), _strides3(k), _p(v), _strides3(v), _p(o), _strides3(o), _p(do), _strides3(do),
                   _p(kt), _p(qt), _p(dot), _p(lse), _p(delta), _p(dq), _strides3(dq), _p(dk), _strides3(dk), _p(dv), _strides3(dv),
                   B, H, Sq, Sk, D, scale, _p(bias_lut), _p(kmask), 1 if causal else 0, sp, site, p, _p(drop_bits), _stream()))


# ------------------------------------------------------------------------------------------------ side kernels
CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)  # blip_processors.py:63-66


def patchify(video: torch.Tensor, out: torch.Tensor, patch: int, mean=CLIP_MEAN, std=CLIP_STD):
    """video: fp32 normalised frames, or uint8 frames (normalisation fused into the load)"""
    F_, _, IMG, _ = video.shape
    f16 = out.dtype == torch.float16   # fp16 patch rows for the fp16-operand ViT
    if video.dtype == torch.uint8:
        _chk((_patchify_u8_f16 if f16 else _patchify_u8)(_p(video), (C.c_float * 3)(*mean), (C.c_float * 3)(*std), _p(out), F_, IMG, patch, out.shape[1], _stream()))
    else:
        _chk((_patchify_f16 if f16 else _patchify)(_p(video), _p(out), F_, IMG, patch, out.shape[1], _stream()))


def vit_assemble(patch, cls, pos, x):
    F_, T, D = x.shape
    _chk(_vit_asm(_p(patch), _p(cls), _p(pos), _p(x), F_, T - 1, D, _stream()))


def row_copy(src, src_idx, dst, dst_idx, accumulate=False):
    n = src_idx.numel()
    _chk(_row_copy(_p(src), _ld(src), _p(src_idx), _p(dst), _ld(dst), _p(dst_idx), n, src.shape[1], 1 if accumulate else 0, _stream()))


def mean_pool(x, out):
    F_, n, D = x.shape
    _chk(_mean_pool(_p(x), _p(out), F_, n, D, _stream()))


def mean_pool_bwd(dout, dx):
    F_, n, D = dx.shape
    _chk(_mean_pool_bwd(_p(dout), _p(dx), F_, n, D, _stream()))


def cast_dropout(x, out_bf16=None, out_f32=None, drop: Optional[Dropout] = None):
    M, N = x.shape
    sp, site, p = _d(drop)
    _chk(_cast_drop(_p(x), _ld(x), _p(out_bf16), _ld(out_bf16), _p(out_f32), _ld(out_f32), M, N, sp, site, p, _stream()))


def gelu_bwd(dy, h, dh):
    _chk(_gelu_bwd(_p(dy), _p(h), _p(dh), dy.numel(), _stream()))


def gated_gelu_bwd(dy, h, dh, drop: Optional[Dropout] = None, dy_ext=None, ext_drop: Optional[Dropout] = None):
    """dy_ext (optional, bf16, laid out like dy): a second part of dy, added under the keep mask of ``ext_drop`` (gemm_ksplit's LoRA part)"""
    M, Nh = dy.shape
    sp, site, p = _d(drop)
    if dy_ext is None:
        _chk(_gated_bwd(_p(dy), _ld(dy), _p(h), _ld(h), _p(dh), _ld(dh), M, Nh, sp, site, p, _stream()))
        return
    esp, esite, ep = _d(ext_drop)
    assert _ld(dy_ext) == _ld(dy) and dy_ext.dtype == dy.dtype and (sp == esp or not sp or not esp)
    _chk(_gated_bwd_parts(_p(dy), _p(dy_ext), _ld(dy), _p(h), _ld(h), _p(dh), _ld(dh), M, Nh, sp or esp, site, p, esite, ep, _stream()))


def sum_parts(parts, out, residual=None):
    """out = (residual) + parts[0] + parts[1] + ... in part order; parts fp32 [n, M, N]"""
    n, M, N = parts.shape
    _req(parts, torch.float32, "sum_parts.parts")
    _req(out, torch.float32, "sum_parts.out")
    assert parts.stride(2) == 1
    _chk(_sum_parts(_p(parts), parts.stride(1), parts.stride(0), n, _p(residual), _ld(residual), _p(out), _ld(out), M, N, _stream()))


def gemm_ksplit(a, w, parts, K: int, k_splits: int, ext=None, tile_cfg: int = 13):
    """parts[s] = a[:, K range s] @ w[:, K range s]^T for s < k_splits; with ext = (g [M, >= 64], acatt [N, >= 64]) parts[k_splits] = g @ acatt^T.
    ``parts``: [k_splits (+ 1), M, N] fp32 or bf16 (rows may be strided).  The hand-pipelined 4-wave kernel in its K-split form
    (mrblip_gemm_ksplit): for outputs with too few 256-wide tiles to fill the chip."""
    M, N = parts.shape[1], parts.shape[2]
    _req(a, torch.bfloat16, "gemm_ksplit.a")
    _req(w, torch.bfloat16, "gemm_ksplit.w")
    assert parts.shape[0] == k_splits + (1 if ext is not None else 0) and parts.stride(2) == 1 and a.shape[0] == M and w.shape[0] == N
    g, at = ext if ext is not None else (None, None)
    _chk(_gemm_ksplit(_p(a), _ld(a), _p(w), _ld(w), _p(g), _ld(g), _p(at), _ld(at), M, N, int(K), _p(parts), parts.stride(1), parts.stride(0),
                      1 if parts.dtype == torch.float32 else 0, int(k_splits), int(tile_cfg), _stream()))


def cross_entropy(logits, labels_i32, inv_count, loss, dlogits=None, n_valid_dev=None):
    """n_valid_dev (int32 device tensor): inv_count = 1 / max(n_valid, 1) is computed on the device instead (inv_count is then ignored)"""
    R, V = logits.shape
    _reduce_ws(logits.device)
    if n_valid_dev is not None:
        _chk(_ce_nv(_p(logits), _ld(logits), _p(labels_i32), R, V, _p(n_valid_dev), _p(loss), _p(dlogits), _ld(dlogits), _stream()))
        return
    _chk(_ce(_p(logits), _ld(logits), _p(labels_i32), R, V, inv_count, _p(loss), _p(dlogits), _ld(dlogits), _stream()))


def adamw(p, g, m, v, hyper, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, guard=None):
    """guard (optional int32 device word, see thin_error_word): the update is applied only while it is zero"""
    if guard is None:
        _chk(_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), beta1, beta2, eps, weight_decay, _stream()))
    else:
        _chk(_adamw_guarded(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), beta1, beta2, eps, weight_decay, _p(guard), _stream()))


def seed_bump(seed):
    _chk(_seed_bump(_p(seed), _stream()))


def gemm_prefetch(t, n_blocks: int = 64, nbytes: int = None, t2=None):
    """The NEXT gemm() / lora_dx() launch of this thread also reads tensor t's bytes (contiguous; or its first nbytes) with n_blocks
    extra workgroups and drops them: a later launch finds them in the memory-side cache.  Not a launch (no _chk count)."""
    rc = _gemm_set_prefetch(_p(t), t.numel() * t.element_size() if nbytes is None else nbytes,
                            _p(t2) if t2 is not None else None, t2.numel() * t2.element_size() if t2 is not None else 0, n_blocks)
    if rc != 0:
        raise MrblipError(_lib.mrblip_last_error().decode())


def prefetch(t, n_blocks: int = 32):
    """Read tensor t's bytes (contiguous) on the current stream and drop them: the next launch finds them in the memory-side cache."""
    assert t.is_contiguous()
    _chk(_prefetch(_p(t), t.numel() * t.element_size(), n_blocks, _stream()))


def lora_dx_add(dx, G, acat, drop: Optional[Dropout] = None):
    """dx[m,k] += mask(m,k) * sum_r G[m,r] * acat[r,k];  acat bf16 [R,K]"""
    M, K = dx.shape
    sp, site, p = _d(drop)
    _chk(_lora_dx(_p(dx), _ld(dx), 1 if dx.dtype == torch.float32 else 0, _p(G), _ld(G), _p(acat), acat.shape[0], M, K, sp, site, p, _stream()))
