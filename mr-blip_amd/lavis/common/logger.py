"""SmoothedValue / MetricLogger as used by the train loop (lavis/common/logger.py:19-195), trimmed to what the path prints."""
import datetime
import logging
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist

from lavis.common import dist_utils


class SmoothedValue:
    def __init__(self, window_size=20, fmt="{median:.4f} ({global_avg:.4f})"):
        self.deque, self.total, self.count, self.fmt = deque(maxlen=window_size), 0.0, 0, fmt

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not dist_utils.is_dist_avail_and_initialized():
            return
        dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def global_avg(self):
        return self.total / max(self.count, 1)

    def __str__(self):
        return self.fmt.format(median=self.median, global_avg=self.global_avg)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters, self.delimiter = defaultdict(SmoothedValue), delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            self.meters[k].update(v.item() if torch.is_tensor(v) else float(v))

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def global_avg(self):
        return self.delimiter.join(f"{n}: {m.global_avg:.4f}" for n, m in self.meters.items())

    def __str__(self):
        return self.delimiter.join(f"{n}: {m}" for n, m in self.meters.items())

    def log_every(self, iterable, print_freq, header=""):
        start = time.time()
        n = len(iterable)
        for i, obj in enumerate(iterable):
            yield obj
            if i % print_freq == 0 or i == n - 1:
                el = time.time() - start
                eta = datetime.timedelta(seconds=int(el / (i + 1) * (n - i - 1)))
                logging.info(f"{header} [{i}/{n}] eta: {eta} {self} time/iter: {el / (i + 1):.4f}")


def setup_logger():
    logging.basicConfig(level=logging.INFO if dist_utils.is_main_process() else logging.WARN, format="%(asctime)s [%(levelname)s] %(message)s")
