"""One process per GPU over torch.distributed; backend "nccl" IS RCCL on ROCm (xGMI inside a node).  Mirrors the helpers of
lavis/common/dist_utils.py:33-114 (rank/world queries, init_distributed_mode from env://, main_process decorator)."""
import datetime
import functools
import os

import torch
import torch.distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def init_distributed_mode(args):
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    else:
        print("Not using distributed mode")
        args.distributed = False
        args.rank, args.world_size, args.gpu = 0, 1, 0
        return
    args.distributed = True
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(args.gpu)
    backend = "nccl" if use_gpu else "gloo"
    args.dist_backend = backend
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    print(f"| distributed init (rank {args.rank}, world {args.world_size}, backend {backend})", flush=True)
    dist.init_process_group(backend=backend, init_method=args.get("dist_url", "env://"), world_size=args.world_size, rank=args.rank,
                            timeout=datetime.timedelta(minutes=30))
    dist.barrier()


def main_process(func):
    @functools.wraps(func)
    def wrapper(*a, **k):
        if is_main_process():
            return func(*a, **k)

    return wrapper
