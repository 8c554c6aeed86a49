"""Does a hipGraph shorten the GPU-side gap between small DEPENDENT kernels?  A chain of N tiny launches of the library's own kernels
(cast_dropout on a decoder-sized [8 x 2048] buffer, ping-pong so each depends on the previous) timed with HIP events:
  (a) eager, host far ahead (a long GEMM is queued first so all N launches sit in the queue before the GPU reaches them),
  (b) the same chain captured once in a torch.cuda.CUDAGraph (= hipGraph) and replayed.
Prints us per launch for both: the difference is what capturing the 12-token decoder (~1700 launches per step) could save."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops  # noqa: E402

dev = torch.device("cuda:0")
N = 1000
a = torch.randn(8, 2048, device=dev)
b = torch.empty(8, 2048, device=dev)
big_a = torch.randn(8192, 8192, device=dev).bfloat16()
big_w = torch.randn(8192, 8192, device=dev).bfloat16()
big_o = torch.empty(8192, 8192, dtype=torch.bfloat16, device=dev)


def chain():
    x, y = a, b
    for _ in range(N):
        ops.cast_dropout(x, out_f32=y)
        x, y = y, x


def timed(fn, pre_queue):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if pre_queue:
        for _ in range(6):
            ops.gemm(big_a, big_w, big_o)   # ~6 x 1 ms of GPU work: the host enqueues the whole chain meanwhile
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / N


st = torch.cuda.Stream()
with torch.cuda.stream(st):
    chain()
    torch.cuda.synchronize()
    t_eager_host_bound = timed(chain, False)
    t_eager = timed(chain, True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        chain()
    g.replay()
    torch.cuda.synchronize()
    t_graph = timed(g.replay, False)
    t_graph2 = timed(g.replay, True)
print("us per dependent tiny launch: eager (host-bound) %.2f | eager (pre-queued) %.2f | hipGraph replay %.2f / %.2f" % (t_eager_host_bound, t_eager, t_graph, t_graph2))
