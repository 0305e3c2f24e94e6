"""Device-side prefetch of preference pairs: DataLoader workers build the paired layout
[B,2,F,C,H,W] on the host (dataset.collate_paired), this iterator pins and copies batch k+1 to HBM on a side stream
while step k computes, so the timed step never waits on PCIe (6.4 MB per pair, ~0.1 ms at Gen5 x16)."""
from typing import Iterable, Iterator

import torch


class PairedPrefetcher:
    def __init__(self, loader: Iterable, device=None, depth: int = 2):
        self.loader = loader
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.depth = max(1, depth)
        self.stream = torch.cuda.Stream(device=self.device)

    def _stage(self, batch):
        out = {}
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                if torch.is_tensor(v):
                    v = v.pin_memory() if not v.is_pinned() else v
                    out[k] = v.to(self.device, non_blocking=True)
                else:
                    out[k] = v
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return out, ev

    def __iter__(self) -> Iterator[dict]:
        it = iter(self.loader)
        queue = []
        try:
            while len(queue) < self.depth:
                queue.append(self._stage(next(it)))
        except StopIteration:
            pass
        while queue:
            batch, ev = queue.pop(0)
            torch.cuda.current_stream(self.device).wait_event(ev)
            for v in batch.values():
                if torch.is_tensor(v):
                    v.record_stream(torch.cuda.current_stream(self.device))
            try:
                queue.append(self._stage(next(it)))
            except StopIteration:
                pass
            yield batch
