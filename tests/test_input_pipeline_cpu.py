"""f2 — input pipeline on the host: frame-index sampling against the reference's load_video (golden: tests/golden/video_sampling.json,
made by make_golden_video.py through a stub decoder), frame dumps -> uint8 [T,3,H,W], the dataset's sample contract, timestamps."""
import json
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))


def test_frame_sampling_matches_reference():
    from lavis.datasets.data_utils import sample_frame_indices

    g = json.load(open(os.path.join(ROOT, "tests", "golden", "video_sampling.json")))["cases"]
    assert len(g) > 150 and sum("error" in c for c in g) < len(g) // 3
    for c in g:
        random.seed(c["seed"])
        if "error" in c:
            with pytest.raises(Exception) as ei:
                sample_frame_indices(c["vlen"], c["fps"], c["n_frms"], c["sampling"], c["clip"], rng=random)
            assert type(ei.value).__name__ == c["error"], c
        else:
            got = sample_frame_indices(c["vlen"], c["fps"], c["n_frms"], c["sampling"], c["clip"], rng=random)
            assert got == c["indices"], c


def test_load_video_from_frame_dump_and_dataset_contract(tmp_path):
    import lavis  # noqa: F401
    from lavis.datasets import MomentRetrievalDataset, collate
    from lavis.datasets.data_utils import load_video

    rs = np.random.RandomState(0)
    frames = rs.randint(0, 256, (90, 32, 32, 3), dtype=np.uint8)
    np.savez(tmp_path / "vid7.npz", frames=frames, fps=np.float64(15.0))
    u8, idx, fps = load_video(str(tmp_path / "vid7.npz"), n_frms=6, height=32, width=32)
    assert u8.dtype == torch.uint8 and tuple(u8.shape) == (6, 3, 32, 32) and fps == 15.0 and idx == [7, 22, 37, 52, 67, 82]
    assert torch.equal(u8[2], torch.from_numpy(frames[37]).permute(2, 0, 1))
    u8r, _, _ = load_video(str(tmp_path / "vid7.npz"), n_frms=6, height=16, width=16)
    assert tuple(u8r.shape) == (6, 3, 16, 16) and u8r.dtype == torch.uint8
    ann = [{"video": "vid7", "qid": 11, "query": "a dog runs", "duration": 6.0, "relevant_windows": [[1, 3]]},
           {"video": "vid7", "qid": 12, "query": "a cat sits", "duration": 6.0, "relevant_windows": [[0, 2], [4, 6]], "start": 1.0, "end": 4.0}]
    json.dump(ann, open(tmp_path / "ann.json", "w"))
    ds = MomentRetrievalDataset(str(tmp_path / "ann.json"), str(tmp_path), n_frms=6, image_size=32)
    s0, s1 = ds[0], ds[1]
    assert set(s0) == {"video", "timestamps", "duration", "query_id", "query_prompt", "task_prompt", "video_prompt_end", "relevant_windows"}
    assert s0["video"].dtype == torch.uint8 and s0["query_prompt"] == "Query: a dog runs\n" and s0["relevant_windows"] == "[[1, 3]]"
    assert s0["timestamps"].tolist() == [round(k / 15.0, 2) for k in [7, 22, 37, 52, 67, 82]] or torch.allclose(s0["timestamps"], torch.tensor([round(k / 15.0, 2) for k in [7, 22, 37, 52, 67, 82]]))
    assert 1.0 <= float(s1["timestamps"][0]) and float(s1["timestamps"][-1]) <= 4.0      # clip proposal (start / end)
    b = collate([s0, s1])
    assert tuple(b["video"].shape) == (2, 6, 3, 32, 32) and b["video"].dtype == torch.uint8 and tuple(b["timestamps"].shape) == (2, 6)
    with pytest.raises(FileNotFoundError):
        load_video(str(tmp_path / "missing.mp4"), n_frms=4)


def test_train_split_random_resized_crop(tmp_path):
    """ADVICE r2 / blip_processors.py:287-312: the train processor crops ONE random region per clip (scale 0.5..1 of the area, aspect
    3/4..4/3), bicubic-resizes it to image_size and shares it across the frames; eval splits are not cropped."""
    import lavis  # noqa: F401
    from lavis.datasets import MomentRetrievalDataset
    from lavis.datasets.data_utils import random_resized_crop_params, random_resized_crop_u8

    rng = random.Random(3)
    for _ in range(300):
        H, W = rng.choice([(224, 224), (180, 320), (256, 144)])
        i, j, h, w = random_resized_crop_params(H, W, scale=(0.5, 1.0), rng=rng)
        assert 0 <= i and i + h <= H and 0 <= j and j + w <= W and h > 0 and w > 0
        assert 0.49 <= h * w / (H * W) <= 1.0 + 1e-9, (h, w, H, W)
        assert 0.73 <= w / h <= 1.37
    # shared across frames: identical input frames give identical output frames, and a gradient image stays monotone after the crop
    ramp = torch.arange(64, dtype=torch.float32).view(1, 1, 1, 64).expand(5, 3, 64, 64).mul(3).to(torch.uint8).contiguous()
    out = random_resized_crop_u8(ramp, 32, rng=random.Random(1))
    assert out.dtype == torch.uint8 and tuple(out.shape) == (5, 3, 32, 32)
    assert all(torch.equal(out[0], out[k]) for k in range(1, 5))
    assert (out[0, 0, 0, 1:].int() - out[0, 0, 0, :-1].int()).min() >= 0
    assert not torch.equal(random_resized_crop_u8(ramp, 32, rng=random.Random(1)), random_resized_crop_u8(ramp, 32, rng=random.Random(2)))
    # dataset: train split (crop_scale) differs from the eval decode of the same clip, eval is the plain resize
    rs = np.random.RandomState(0)
    np.savez(tmp_path / "v.npz", frames=rs.randint(0, 256, (40, 48, 48, 3), dtype=np.uint8), fps=np.float64(10.0))
    json.dump([{"video": "v", "qid": 1, "query": "q", "duration": 4.0, "relevant_windows": [[1, 2]]}], open(tmp_path / "a.json", "w"))
    ev = MomentRetrievalDataset(str(tmp_path / "a.json"), str(tmp_path), n_frms=4, image_size=32)
    tr = MomentRetrievalDataset(str(tmp_path / "a.json"), str(tmp_path), n_frms=4, image_size=32, crop_scale=(0.5, 1.0))
    random.seed(5)
    a, b = ev[0]["video"], tr[0]["video"]
    assert a.dtype == b.dtype == torch.uint8 and a.shape == b.shape == (4, 3, 32, 32) and not torch.equal(a, b)
    assert torch.equal(ev[0]["video"], a)


@pytest.mark.parametrize("name,builder,n_frms,bs,accum,world", [("charades", "charades_sta", 20, 8, 1, 4), ("anet", "anet", 60, 1, 4, 8), ("qvh", "qvh", 60, 1, 8, 8)])
def test_project_configs_of_every_benchmark_dataset(tmp_path, name, builder, n_frms, bs, accum, world):
    """BASELINE.json configs 2-5 through the real entry point: lavis/projects/mr_BLIP/{train,eval}/{qvh,charades,anet}.yaml resolve (builder
    registered, dataset defaults merged, the reference's run values), and the builder gives the TRAIN split the train processor's crop."""
    import argparse

    import lavis  # noqa: F401
    from lavis.common.config import Config
    from lavis.common.registry import registry

    rs = np.random.RandomState(1)
    np.savez(tmp_path / "vid.npz", frames=rs.randint(0, 256, (70, 40, 40, 3), dtype=np.uint8), fps=np.float64(7.0))
    ann = [{"video": "vid", "qid": k, "query": "someone opens a door", "duration": 10.0, "relevant_windows": [[2, 5]]} for k in range(3)]
    for split in ("train", "val", "test"):
        json.dump(ann, open(tmp_path / f"{split}.json", "w"))
    for mode in ("train", "eval"):
        path = os.path.join(ROOT, "mr-blip_amd", "lavis", "projects", "mr_BLIP", mode, name + ".yaml")
        opts = [f"datasets.{builder}.build_info.annotations.{s}.storage={tmp_path}/{s}.json" for s in ("train", "val", "test")]
        opts += [f"datasets.{builder}.build_info.videos.storage={tmp_path}"]
        cfg = Config(argparse.Namespace(cfg_path=path, options=opts))
        assert cfg.model_cfg.arch == "blip2_mr" and cfg.model_cfg.task == "qformer_freeze_lora" and cfg.model_cfg.freeze_vit is True
        assert list(cfg.datasets_cfg) == [builder]
        dcfg = cfg.datasets_cfg[builder]
        assert dcfg.vis_processor.eval.n_frms == n_frms and dcfg.build_info.videos.storage == str(tmp_path)
        if mode == "train":
            r = cfg.run_cfg
            assert (r.batch_size_train, r.accum_grad_iters, r.world_size, r.init_lr, r.weight_decay, r.num_beams) == (bs, accum, world, 3e-4, 0.05, 5)
            assert dcfg.vis_processor.train.name == "blip2_video_train" and dcfg.vis_processor.train.n_frms == n_frms
        dcfg.vis_processor.eval.image_size = 32     # (decode small: this test is about plumbing)
        if "train" in dcfg.vis_processor:
            dcfg.vis_processor.train.image_size = 32
        ds = registry.get_builder_class(builder)(dcfg).build_datasets()
        assert set(ds) == {"train", "val", "test"} and len(ds["train"]) == 3
        if mode == "train":
            assert ds["train"].crop_scale == (0.5, 1.0) and ds["train"].sampling == "random"
        assert ds["val"].crop_scale is None and ds["val"].sampling == "uniform" and ds["test"].T == n_frms
        s = ds["val"][0]
        assert tuple(s["video"].shape) == (n_frms, 3, 32, 32) and s["video"].dtype == torch.uint8
