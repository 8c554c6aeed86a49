"""Dataset builders of the Mr. BLIP path: the annotation-JSON moment-retrieval dataset (qvh / charades_sta / anet) and a synthetic
one with the same sample dict for runs without the video corpora."""
import json
import os

import torch
from torch.utils.data import Dataset

from lavis.common.registry import registry

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
TASK_PROMPT = "Given the video and the query, find the relevant windows.\nRelevant windows: "


def collate(batch):
    out = {}
    for k in batch[0]:
        v = [b[k] for b in batch]
        out[k] = torch.stack(v) if torch.is_tensor(v[0]) else (torch.tensor(v) if isinstance(v[0], float) else v)
    return out


class _MRBase(Dataset):
    collater = staticmethod(collate)

    def _sample(self, video, timestamps, duration, query, windows, qid):
        """the ``samples`` contract of lavis/datasets/datasets/moment_retrieval_dataset.py:17-60"""
        return {"video": video, "timestamps": timestamps, "duration": float(duration), "query_id": qid,
                "query_prompt": "Query: " + query + "\n", "task_prompt": TASK_PROMPT, "video_prompt_end": "<extra_id_0>",
                "relevant_windows": str(windows)}


class SyntheticMomentRetrievalDataset(_MRBase):
    def __init__(self, n_items=16, n_frms=60, image_size=224, duration=150.0, seed=0):
        self.n, self.T, self.img, self.dur, self.seed = n_items, n_frms, image_size, duration, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        u8 = torch.randint(0, 256, (self.T, 3, self.img, self.img), generator=g, dtype=torch.uint8)
        mean, std = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1), torch.tensor(CLIP_STD).view(1, 3, 1, 1)
        video = (u8.float() / 255.0 - mean) / std
        ts = torch.tensor([round((k + 0.5) * self.dur / self.T, 2) for k in range(self.T)], dtype=torch.float32)
        s = int(torch.randint(0, int(self.dur) - 10, (1,), generator=g))
        return self._sample(video, ts, self.dur, "a person opens the red door and walks into the kitchen", [[s, s + 8]], f"syn{i}")


def default_frame_loader(path, n_frms, size, sampling="uniform", clip_proposal=None):
    """decord / av / torchvision (whichever is installed) or a .npy / .npz frame dump -> (uint8 [T,3,size,size], indices, fps)"""
    from lavis.datasets.data_utils import load_video

    return load_video(path, n_frms=n_frms, height=size, width=size, sampling=sampling, clip_proposal=clip_proposal)


class MomentRetrievalDataset(_MRBase):
    """annotation JSON of {video, qid, query, duration, relevant_windows[, start, end]} (moment_retrieval_dataset.py:17-60).  Frames are
    decoded by ``frame_loader`` (default: lavis.datasets.data_utils.load_video) and handed on as **uint8** — the processor's
    ToTensor + Normalize is fused into the patch-embed load on the GPU (a quarter of the H2D bytes).

    ``crop_scale`` = (min_scale, max_scale) switches on the train processor's augmentation (Blip2VideoTrainProcessor,
    blip_processors.py:287-312): one RandomResizedCrop per clip, bicubic, shared by all frames; None (eval splits) = frames as decoded."""

    def __init__(self, ann_path, vis_root, n_frms=60, image_size=224, frame_loader=None, sampling="uniform", video_ext=".mp4", crop_scale=None):
        self.ann = json.load(open(ann_path))
        self.vis_root, self.T, self.img = vis_root, n_frms, image_size
        self.frame_loader = frame_loader or default_frame_loader
        self.sampling, self.ext = sampling, video_ext
        self.crop_scale = tuple(float(x) for x in crop_scale) if crop_scale else None

    def __len__(self):
        return len(self.ann)

    def _path(self, name):
        p = os.path.join(self.vis_root, name)
        if os.path.exists(p):
            return p
        for ext in (self.ext, ".npz", ".npy"):
            if os.path.exists(p + ext):
                return p + ext
        return p + self.ext   # (the reference appends ".mp4" unconditionally: moment_retrieval_dataset.py:30)

    def __getitem__(self, i):
        a = self.ann[i]
        clip = [float(a["start"]), float(a["end"])] if "start" in a else None
        try:
            u8, idx, fps = self.frame_loader(self._path(a["video"]), self.T, self.img, sampling=self.sampling, clip_proposal=clip)
        except TypeError:  # a user-supplied loader with the short signature
            u8, idx, fps = self.frame_loader(self._path(a["video"]), self.T, self.img)
        if self.crop_scale is not None:
            from lavis.datasets.data_utils import random_resized_crop_u8
            u8 = random_resized_crop_u8(u8, self.img, scale=self.crop_scale)
        ts = torch.tensor([round(float(k / fps), 2) for k in idx])
        return self._sample(u8, ts, a["duration"], a["query"], a["relevant_windows"], a["qid"])


class _Builder:
    DATASET_CONFIG_DICT = {}

    def __init__(self, cfg):
        self.config = cfg

    @classmethod
    def default_config_path(cls, type="default"):
        return os.path.join(registry.get_path("library_root"), cls.DATASET_CONFIG_DICT[type])


@registry.register_builder("synthetic_mr")
class SyntheticBuilder(_Builder):
    DATASET_CONFIG_DICT = {"default": "configs/datasets/qvh/synthetic.yaml"}

    def build_datasets(self):
        c = self.config
        vp = c.get("vis_processor", {}).get("train", {})
        kw = dict(n_frms=vp.get("n_frms", 60), image_size=vp.get("image_size", 224), duration=c.get("duration", 150.0))
        return {"train": SyntheticMomentRetrievalDataset(n_items=c.get("n_train", 16), seed=1, **kw),
                "val": SyntheticMomentRetrievalDataset(n_items=c.get("n_val", 4), seed=2, **kw)}


class MomentRetrievalBuilder(_Builder):
    """lavis/datasets/builders/moment_retrieval_builder.py:16-23 + base_dataset_builder.py: one MomentRetrievalDataset per annotation
    split; the TRAIN split gets the train processor (random frame per interval + one RandomResizedCrop per clip, scale
    [min_scale, max_scale] = [0.5, 1.0] unless the vis_processor config says otherwise), every other split the eval processor
    (middle frame per interval, no crop).  Splits whose annotation file does not exist are skipped with a warning."""

    def build_datasets(self):
        info = self.config.build_info
        vps = self.config.get("vis_processor", {})
        out = {}
        for split, ann in info.annotations.items():
            if not os.path.isfile(str(ann.storage)):
                continue
            is_train = split == "train"
            vp = vps.get("train" if is_train else "eval", None) or vps.get("train", None) or vps.get("eval", None) or {}
            crop = (vp.get("min_scale", 0.5), vp.get("max_scale", 1.0)) if is_train and vp.get("name", "blip2_video_train") == "blip2_video_train" else None
            out[split] = MomentRetrievalDataset(ann.storage, info.videos.storage, n_frms=vp.get("n_frms", 60), image_size=vp.get("image_size", 224),
                                                sampling="random" if is_train else "uniform", crop_scale=crop)
        if not out:
            import logging
            logging.warning("%s builder: none of the annotation files %s exists — no dataset was built", type(self).__name__,
                            {k: str(v.storage) for k, v in info.annotations.items()})
        return out


@registry.register_builder("qvh")
class QVHBuilder(MomentRetrievalBuilder):
    DATASET_CONFIG_DICT = {"default": "configs/datasets/qvh/defaults.yaml"}


@registry.register_builder("charades_sta")
class Charades_STABuilder(MomentRetrievalBuilder):   # moment_retrieval_builder.py:51-55
    DATASET_CONFIG_DICT = {"default": "configs/datasets/charades_sta/defaults.yaml"}


@registry.register_builder("anet")
class ANetBuilder(MomentRetrievalBuilder):           # moment_retrieval_builder.py:79-83
    DATASET_CONFIG_DICT = {"default": "configs/datasets/anet/defaults.yaml"}
