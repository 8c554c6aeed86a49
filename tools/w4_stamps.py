"""Where does a 256x256 tile of the 4-wave ViT GEMM spend its time?  Needs the instrumented build
    tools/exp_build.sh stamps gemm.hip -DEXP_W4_STAMPS      and      MRBLIP_LIB=exp_libs/lib_stamps.so python tools/w4_stamps.py
Every block stamps (s_memrealtime, 10 ns) tile start / K-loop start / K-loop end / epilogue end of each tile it works on; printed: the
launch's HIP-event time and, per round, the mean over blocks of  prologue (open -> first K-tile landed), K loop, epilogue."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    shapes = [("vit_fc1 bias+GELU", 15420, 6144, 1408, 1, False), ("vit_qkv bias", 15420, 4224, 1408, 0, False), ("vit_fc2 bias+res f32", 15420, 1408, 6144, 0, True)]
    lib = ops._lib
    lib.mrblip_debug_w4_stamps.argtypes = [C.c_void_p]
    out_rows = []
    for name, M, N, K, act, res in shapes:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, dtype=torch.float32 if res else torch.bfloat16, device=dev)
        resid = torch.randn(M, N, device=dev) if res else None
        f = lambda: ops.gemm(a, w, out, bias=bias, act=act, residual=resid, tile_cfg=13)  # noqa: E731
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            f()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 100.0
        buf = np.zeros(256 * 8 * 4, dtype=np.uint64)
        assert lib.mrblip_debug_w4_stamps(buf.ctypes.data) == 0
        st = buf.reshape(256, 8, 4).astype(np.int64)
        tiles = (M + 255) // 256 * ((N + 255) // 256)
        t0 = st[:, 0, 0].min()
        row = dict(shape=name, M=M, N=N, K=K, launch_us=round(us, 1), TF=round(2.0 * M * N * K / us / 1e6, 1), tiles=tiles, rounds=round(tiles / 256, 2), per_round=[])
        for r in range(8):
            live = np.arange(256) + r * 256 < tiles  # (approximately: the XCD-wise walk gives some blocks one tile more)
            v = st[:, r, :]
            ok = (v[:, 3] > v[:, 0]) & (v[:, 0] >= t0)
            if ok.sum() == 0:
                break
            d = v[ok]
            row["per_round"].append(dict(round=r, blocks=int(ok.sum()), start_us=round(float((d[:, 0] - t0).mean()) / 100, 2),
                                         prologue_us=round(float((d[:, 1] - d[:, 0]).mean()) / 100, 2), kloop_us=round(float((d[:, 2] - d[:, 1]).mean()) / 100, 2),
                                         epilogue_us=round(float((d[:, 3] - d[:, 2]).mean()) / 100, 2), end_us=round(float((d[:, 3] - t0).max()) / 100, 2)))
        print(json.dumps(row), flush=True)
        out_rows.append(row)
    json.dump(out_rows, open(os.path.join(ROOT, "gpurun_out", "w4_stamps.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
