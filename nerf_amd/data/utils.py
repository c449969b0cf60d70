"""DataLoader factory (reference nerf/data/utils.py:5-48): same arguments, same three
loaders (train shuffled), so `fit` batches rays exactly as the reference does."""
from typing import Optional, Tuple

from torch.utils.data import DataLoader, Dataset


def loaders(train_data: Optional[Dataset], val_data: Optional[Dataset], test_data: Optional[Dataset],
            batch_size: int, jobs: int) -> Tuple[Optional[DataLoader], Optional[DataLoader], Optional[DataLoader]]:
    def make(data, shuffle):
        if not data:
            return None
        return DataLoader(data, batch_size=batch_size, shuffle=shuffle, num_workers=jobs, pin_memory=True)

    return make(train_data, True), make(val_data, False), make(test_data, False)
