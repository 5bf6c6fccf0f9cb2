from .OCR import OCRNet  # noqa: F401
from .DeepLabv3Plus import DeepLabv3Plus  # noqa: F401
