"""Task tables of the CaDIS benchmark as the reference uses them (data restated from
utils/defaults.py:16-33,112-237 of the reference: number of network classes per experiment,
ignore label, category index lists used for the mIoU break-down)."""

# experiment -> number of classes the network predicts (models/OCR.py:41-42)
NUM_CLASSES = {1: 8, 2: 17, 3: 25}
# label value that marks "ignore" after remapping (utils/utils.py:45-46); None = no ignore label
IGNORE_LABEL = {1: None, 2: 17, 3: 25}

CATEGORIES = {
    1: {"anatomies": [0, 4, 5, 6], "instruments": [7], "others": [1, 2, 3], "rare": [2]},
    2: {"anatomies": [0, 4, 5, 6], "instruments": [7, 8, 9, 10, 11, 12, 13, 14, 15, 16], "others": [1, 2, 3],
        "rare": [16, 10, 9, 12, 14]},
    3: {"anatomies": [0, 4, 5, 6], "instruments": list(range(7, 25)), "others": [1, 2, 3],
        "rare": [24, 20, 21, 22, 18, 23, 19, 16, 12, 11, 14]},
}


def num_classes(experiment):
    return NUM_CLASSES[experiment]


def ce_ignore_index(experiment):
    """losses/LossWrapper.py:17-24"""
    return {2: 17, 3: 25}.get(experiment, -100)
