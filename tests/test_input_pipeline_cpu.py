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
