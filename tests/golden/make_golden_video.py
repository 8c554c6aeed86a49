"""Golden for the input pipeline's frame sampling: the REFERENCE's ``load_video`` (lavis/datasets/data_utils.py:30-85) driven through a stub
``decord.VideoReader`` (no codec in the build image) for a grid of (length, fps, n_frms, clip_proposal, sampling) cases.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_video.py  -> video_sampling.json"""
import json
import os
import random
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MRBLIP_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
STATE = {}


class VideoReader:
    def __init__(self, uri, height=-1, width=-1):
        self.n, self.fps = STATE["vlen"], STATE["fps"]

    def __len__(self):
        return self.n

    def get_avg_fps(self):
        return self.fps

    def get_batch(self, indices):
        STATE["asked"] = [int(i) for i in indices]
        return torch.zeros(len(indices), 2, 2, 3)


decord = types.ModuleType("decord")
decord.VideoReader = VideoReader
decord.bridge = types.SimpleNamespace(set_bridge=lambda *_: None)
sys.modules["decord"] = decord
for name in ("tqdm", "iopath", "iopath.common", "iopath.common.file_io", "webdataset", "torchvision", "torchvision.datasets", "torchvision.datasets.utils"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["tqdm"].tqdm = lambda x, *a, **k: x
sys.modules["iopath.common.file_io"].g_pathmgr = None
for n in ("check_integrity", "download_file_from_google_drive", "extract_archive"):
    setattr(sys.modules["torchvision.datasets.utils"], n, lambda *a, **k: None)
for pkg in ("lavis", "lavis.common", "lavis.datasets", "lavis.datasets.datasets"):
    m = types.ModuleType(pkg)
    m.__path__ = [os.path.join(REF, *pkg.split("."))]
    sys.modules[pkg] = m
reg = types.ModuleType("lavis.common.registry")
reg.registry = types.SimpleNamespace(get=lambda k: int(1e10))
sys.modules["lavis.common.registry"] = reg
ds = types.ModuleType("lavis.datasets.datasets.base_dataset")
ds.ConcatDataset = object
sys.modules["lavis.datasets.datasets.base_dataset"] = ds
import importlib  # noqa: E402

du = importlib.import_module("lavis.datasets.data_utils")


def main():
    cases = []
    for vlen, fps in ((4500, 30.0), (3597, 23.976), (37, 12.5), (10, 25.0), (150, 1.0)):
        for n_frms in (4, 20, 60, 120):
            for clip in (None, [2.0, 9.5], [0.0, 1e6], [-3.0, 4.0]):
                for sampling in ("uniform", "random", "headtail"):
                    if sampling == "headtail" and (vlen // 2 < min(n_frms, vlen) // 2):
                        continue
                    STATE.update(vlen=vlen, fps=fps)
                    random.seed(1000 + len(cases))
                    seed = 1000 + len(cases)
                    try:
                        frms, indices, f = du.load_video("x.mp4", n_frms=n_frms, height=224, width=224, sampling=sampling, clip_proposal=clip)
                        cases.append(dict(vlen=vlen, fps=fps, n_frms=n_frms, clip=clip, sampling=sampling, seed=seed,
                                          indices=[int(i) for i in indices], shape=list(frms.shape)))
                    except Exception as e:  # noqa: BLE001  (e.g. a clip proposal that starts beyond the video: empty ranges)
                        cases.append(dict(vlen=vlen, fps=fps, n_frms=n_frms, clip=clip, sampling=sampling, seed=seed, error=type(e).__name__))
    json.dump({"cases": cases}, open(os.path.join(HERE, "video_sampling.json"), "w"))
    print("wrote video_sampling.json:", len(cases), "cases")


if __name__ == "__main__":
    main()
