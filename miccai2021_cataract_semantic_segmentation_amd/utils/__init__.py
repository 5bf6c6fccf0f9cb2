from .classes import CATEGORIES, CLASS_REMAP, IGNORE_LABEL, NUM_CLASSES, ce_ignore_index, num_classes  # noqa: F401
from .augment import GpuAugment, gaussian_box_params, sample_blur, sample_color_jitter  # noqa: F401
from .ingest import GpuIngest, remap_lut, sample_flips  # noqa: F401
from .lr_functions import LRFcts  # noqa: F401
from .sampling import RepeatFactorSampler, class_repeat_factors, image_repeat_factors  # noqa: F401


def __getattr__(name):  # metrics need the HIP library: import lazily so CPU-only tooling can use the rest
    if name in ("t_get_confusion_matrix", "t_get_pixel_accuracy", "t_get_miou", "t_get_mean_iou"):
        from . import metrics
        return getattr(metrics, name)
    if name == "PinnedFrameLoader":
        from .loader import PinnedFrameLoader
        return PinnedFrameLoader
    raise AttributeError(name)
