from .OCR import OCRNet  # noqa: F401
from .DeepLabv3Plus import DeepLabv3, DeepLabv3Plus  # noqa: F401
from .HRNetv2 import HRNetv2  # noqa: F401
from .EncDec import EncDec, UPerNet  # noqa: F401
from .FCN import FCN  # noqa: F401
from . import backbone  # noqa: F401
