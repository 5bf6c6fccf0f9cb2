from .lovasz import LovaszSoftmax  # noqa: F401
from .cross_entropy import CrossEntropyLoss  # noqa: F401
from .ohem import OhemCrossEntropy  # noqa: F401
from .two_scale import TwoScaleLoss  # noqa: F401
from .wrapper import LossWrapper  # noqa: F401
