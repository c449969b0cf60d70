from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass
class History:
    """What `fit` returns (reference nerf/utils/history.py): per-epoch (mse, psnr)
    for train and val, the test pair, and every step's learning rate."""
    train: List[Tuple[float, float]] = field(default_factory=list)
    val: List[Tuple[float, float]] = field(default_factory=list)
    test: Tuple[float, float] = None
    lr: List[float] = field(default_factory=list)
