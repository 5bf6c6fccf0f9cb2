from .base import BaseManager, merge_defaults  # noqa: F401
from .segmentation import (DeepLabv3Manager, DeepLabv3PlusManager, EncDecManager, FCNManager, HRNetv2Manager, OCRNetManager,  # noqa: F401
                           SyntheticCataractDataset)
