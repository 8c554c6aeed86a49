"""``PrefetchLoader``: the next batch's host->device copy runs on a side HIP stream while the current step computes (behaviour of
lavis/datasets/datasets/dataloader_utils.py:46-125).  With uint8 frames a QVH clip is 9 MB (36 MB as the reference's fp32 tensor)."""
import torch


def move_to_cuda(batch, device=None):
    if torch.is_tensor(batch):
        return (batch if batch.is_pinned() else batch.pin_memory()).to(device or "cuda", non_blocking=True) if not batch.is_cuda else batch
    if isinstance(batch, dict):
        return {k: move_to_cuda(v, device) for k, v in batch.items()}
    if isinstance(batch, (list, tuple)):
        return type(batch)(move_to_cuda(v, device) for v in batch)
    return batch


def _record_stream(batch, stream):
    if torch.is_tensor(batch):
        if batch.is_cuda:
            batch.record_stream(stream)
    elif isinstance(batch, dict):
        for v in batch.values():
            _record_stream(v, stream)
    elif isinstance(batch, (list, tuple)):
        for v in batch:
            _record_stream(v, stream)


class PrefetchLoader:
    """only the frame tensor ("video") goes to the device ahead of time; the small fp32 timestamp / duration tensors stay on the host,
    where the prompt layout is built from them"""

    def __init__(self, loader, device=None, keys=("video",)):
        self.loader, self.device, self.keys = loader, device, tuple(keys)
        self.stream = torch.cuda.Stream(device=device)

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):
        return getattr(self.loader, name)

    def _preload(self, it):
        try:
            batch = next(it)
        except StopIteration:
            self.batch = None
            return
        with torch.cuda.stream(self.stream):
            if isinstance(batch, dict):
                batch = {k: (move_to_cuda(v, self.device) if k in self.keys else v) for k, v in batch.items()}
            else:
                batch = move_to_cuda(batch, self.device)
        self.batch = batch

    def __iter__(self):
        it = iter(self.loader)
        self._preload(it)
        while self.batch is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
            batch = self.batch
            _record_stream(batch, torch.cuda.current_stream())
            self._preload(it)
            yield batch
