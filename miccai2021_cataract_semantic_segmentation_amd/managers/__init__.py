from .base import BaseManager, merge_defaults  # noqa: F401
from .segmentation import DeepLabv3PlusManager, HRNetv2Manager, OCRNetManager, SyntheticCataractDataset  # noqa: F401
