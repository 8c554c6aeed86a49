"""Frame sampling + decoding for the hot path's input.  Behaviour of ``load_video`` in the reference's
lavis/datasets/data_utils.py:30-85: split [start, end) into n_frms equal intervals (np.linspace(...).astype(int)), take the middle
frame of each ("uniform"), a random frame of each ("random") or random head/tail halves ("headtail"); returns (frames, indices, fps).

Difference by design: frames stay **uint8** ``[T, 3, H, W]`` — a quarter of the H2D bytes of the reference's fp32 tensor — and
ToTensor + Normalize(CLIP mean/std) run fused inside the patch-embed load on the GPU (csrc/elementwise.hip ``patchify_u8``,
bit-identical to the processor's arithmetic: tests/test_model_gpu.py::test_uint8_frames_equal_normalised_frames).

Decoders are optional (none of decord / av / torchvision is in the build image): the first importable one is used; ``.npy`` / ``.npz``
frame dumps (uint8 [N, H, W, 3] + fps) are always readable so the whole path can be exercised without a codec.
"""
import os
import random as rnd

import numpy as np
import torch

MAX_INT = int(1e10)  # registry.get("MAX_INT") of the reference


def sample_frame_indices(vlen: int, fps: float, n_frms: int, sampling: str = "uniform", clip_proposal=None, rng=rnd):
    """indices of the frames load_video decodes (data_utils.py:42-79), as a pure function of the container's length and frame rate"""
    n_frms = min(n_frms, vlen)
    if clip_proposal is None:
        start, end = 0, vlen
    else:
        start, end = max(int(clip_proposal[0] * fps), 0), min(int(clip_proposal[1] * fps), vlen)
    edges = np.linspace(start=start, stop=end, num=n_frms + 1).astype(int)
    spans = list(zip(edges[:-1].tolist(), edges[1:].tolist()))
    if sampling == "uniform":
        indices = [min((a + b) // 2, vlen - 1) for a, b in spans]
    elif sampling == "random":
        indices = [a if a == b else rng.choice(range(a, b)) for a, b in spans]
    elif sampling == "headtail":
        indices = sorted(rng.sample(range(vlen // 2), n_frms // 2)) + sorted(rng.sample(range(vlen // 2, vlen), n_frms // 2))
    else:
        raise NotImplementedError(sampling)
    if len(indices) < n_frms:
        indices = indices + [indices[-1]] * (n_frms - len(indices))
    return indices


class _ArrayReader:
    """``.npy`` (uint8 [N,H,W,3], 30 fps assumed) or ``.npz`` (frames=..., fps=...) frame dump"""

    def __init__(self, path):
        if path.endswith(".npz"):
            z = np.load(path)
            self.frames, self.fps = z["frames"], float(z["fps"])
        else:
            self.frames, self.fps = np.load(path, mmap_mode="r"), 30.0

    def __len__(self):
        return int(self.frames.shape[0])

    def get(self, indices, height, width):
        f = torch.from_numpy(np.ascontiguousarray(self.frames[indices])).permute(0, 3, 1, 2)  # T,3,H,W uint8
        if height > 0 and width > 0 and (f.shape[-2] != height or f.shape[-1] != width):
            f = torch.nn.functional.interpolate(f.float(), size=(height, width), mode="bilinear", align_corners=False).round().clamp(0, 255).to(torch.uint8)
        return f


def _open(path, height, width):
    if path.endswith((".npy", ".npz")):
        r = _ArrayReader(path)
        return len(r), r.fps, lambda idx: r.get(idx, height, width)
    try:
        import decord  # type: ignore

        vr = decord.VideoReader(uri=path, height=height, width=width)
        return len(vr), float(vr.get_avg_fps()), lambda idx: torch.from_numpy(vr.get_batch(idx).asnumpy()).permute(0, 3, 1, 2).contiguous()
    except ImportError:
        pass
    try:
        import av  # type: ignore

        c = av.open(path)
        st = c.streams.video[0]
        frames = [f.to_ndarray(format="rgb24") for f in c.decode(st)]
        fps = float(st.average_rate)
        arr = np.stack(frames)
        r = _ArrayReader.__new__(_ArrayReader)
        r.frames, r.fps = arr, fps
        return len(frames), fps, lambda idx: r.get(idx, height, width)
    except ImportError:
        pass
    try:
        from torchvision.io import read_video  # type: ignore

        v, _, info = read_video(path, pts_unit="sec", output_format="THWC")
        r = _ArrayReader.__new__(_ArrayReader)
        r.frames, r.fps = v.numpy(), float(info["video_fps"])
        return int(v.shape[0]), r.fps, lambda idx: r.get(idx, height, width)
    except ImportError:
        pass
    raise RuntimeError(f"cannot decode {path}: none of decord / av / torchvision is installed (frame dumps .npy / .npz are always readable)")


def load_video(video_path, n_frms=MAX_INT, height=-1, width=-1, sampling="uniform", clip_proposal=None):
    """-> (uint8 frames [T, 3, H, W], frame indices, fps).  Same sampling as the reference; the frames are NOT converted to float."""
    if not os.path.exists(video_path):
        raise FileNotFoundError(video_path)
    vlen, fps, get = _open(video_path, height, width)
    indices = sample_frame_indices(vlen, fps, n_frms, sampling, clip_proposal)
    return get(indices), indices, fps


# ---- train-split augmentation of the reference's "blip2_video_train" processor (lavis/processors/blip_processors.py:287-312):
# transforms_video.RandomResizedCropVideo(image_size, scale=(min_scale, max_scale), interpolation_mode="bicubic") — ONE crop per clip,
# shared by all frames — then ToUint8 -> ToTensorVideo -> Normalize.  The crop + bicubic resize run here on the decoded frames; the
# /255 + Normalize stay fused in the patch-embed load on the GPU (uint8 hand-over).
def random_resized_crop_params(height: int, width: int, scale=(0.5, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), rng=rnd):
    """(i, j, h, w) of torchvision's RandomResizedCrop.get_params, which the reference's RandomResizedCropVideo inherits
    (lavis/processors/transforms_video.py:53-83): ten attempts at area ~ U(scale) * H * W, log-uniform aspect ratio; else the central
    crop with the ratio clamped into ``ratio``."""
    import math

    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * rng.uniform(scale[0], scale[1])
        aspect = math.exp(rng.uniform(log_ratio[0], log_ratio[1]))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = rng.randint(0, height - h)
            j = rng.randint(0, width - w)
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def random_resized_crop_u8(frames: torch.Tensor, size: int, scale=(0.5, 1.0), rng=rnd) -> torch.Tensor:
    """uint8 [T, 3, H, W] -> uint8 [T, 3, size, size]: crop (i, j, h, w) of every frame, bicubic resize (align_corners=False, as
    functional_video.resize), truncation to uint8 (the reference's ToUint8 is ``tensor.to(torch.uint8)``: truncation toward zero).
    Deviation, deliberate: bicubic overshoot outside [0, 255] is clamped — the reference's bare cast wraps it around (a white pixel
    overshooting to 256.3 becomes 0), which is an upstream accident, not a recipe."""
    T, C, H, W = frames.shape
    i, j, h, w = random_resized_crop_params(H, W, scale=scale, rng=rng)
    crop = frames[:, :, i: i + h, j: j + w].float()
    out = torch.nn.functional.interpolate(crop, size=(size, size), mode="bicubic", align_corners=False)
    return out.clamp_(0, 255).to(torch.uint8)
