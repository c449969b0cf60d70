from .features import FeatureMapping, FourierFeatures, PositionalEncoding
from .model import NeRF
from .renderer import BoundedVolumeRaymarcher
from .scheduler import IndendityScheduler, LogDecayScheduler, MipNeRFScheduler, Scheduler

__all__ = ["FeatureMapping", "FourierFeatures", "PositionalEncoding", "NeRF", "BoundedVolumeRaymarcher",
           "IndendityScheduler", "LogDecayScheduler", "MipNeRFScheduler", "Scheduler"]
