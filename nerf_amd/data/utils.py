"""Batch sources for `fit` (reference nerf/data/utils.py:5-48).

`loaders` keeps the reference's arguments and returns its three DataLoaders (train shuffled).
Given a `device`, datasets that hold their rays as three `(N, 3)` tensors `C`, `ro`, `rd` (as the
reference's BlenderDataset does, blender.py:180-182) are instead served by `DeviceRayLoader`:
the arrays live in HBM once and every batch is one gather kernel, with no worker processes,
no per-item `__getitem__` and no host-to-device copy per step."""
from typing import Optional, Tuple

import torch
from torch.utils.data import DataLoader, Dataset

from .. import ops


class DeviceRayLoader:
    """Iterates (C, ro, rd) batches of device-resident rays; a fresh permutation per epoch when
    `shuffle` (drawn on the device from torch's generator, like `DataLoader(shuffle=True)` draws
    its own on the host).  `len()` and the ragged last batch are the DataLoader's."""

    def __init__(self, data, batch_size: int, shuffle: bool, device):
        self.C, self.ro, self.rd = (getattr(data, k).to(device=device, dtype=torch.float32).contiguous()
                                    for k in ("C", "ro", "rd"))
        assert self.C.shape == self.ro.shape == self.rd.shape and self.C.dim() == 2 and self.C.shape[1] == 3
        self.batch_size, self.shuffle, self.device = int(batch_size), shuffle, self.C.device
        self.dataset = data

    def __len__(self) -> int:
        return (self.C.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.C.shape[0]
        perm = torch.randperm(n, device=self.device) if self.shuffle else None
        for s in range(0, n, self.batch_size):
            yield tuple(ops.gather_rays(self.C, self.ro, self.rd, perm, s, min(self.batch_size, n - s)))


def _device_resident(data) -> bool:
    return all(isinstance(getattr(data, k, None), torch.Tensor) for k in ("C", "ro", "rd"))


def loaders(train_data: Optional[Dataset], val_data: Optional[Dataset], test_data: Optional[Dataset],
            batch_size: int, jobs: int, device=None) -> Tuple[Optional[DataLoader], Optional[DataLoader], Optional[DataLoader]]:
    def make(data, shuffle):
        if not data:
            return None
        if device is not None and torch.device(device).type == "cuda" and _device_resident(data):
            return DeviceRayLoader(data, batch_size, shuffle, device)
        return DataLoader(data, batch_size=batch_size, shuffle=shuffle, num_workers=jobs, pin_memory=True)

    return make(train_data, True), make(val_data, False), make(test_data, False)
