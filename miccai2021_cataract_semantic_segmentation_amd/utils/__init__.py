from .classes import CATEGORIES, IGNORE_LABEL, NUM_CLASSES, ce_ignore_index, num_classes  # noqa: F401
