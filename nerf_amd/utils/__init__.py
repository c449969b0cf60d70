from .history import History
from .pbar import tqdm

__all__ = ["History", "tqdm"]
