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


# experiment -> {network class: [raw CaDIS ids]} (data restated from utils/defaults.py of the reference, CLASS_INFO[exp][0];
# key 255 = ignored raw ids, which remap_mask(to_network=True) sends to the ignore label)
CLASS_REMAP = {
    1: {0: [0], 1: [1], 2: [2], 3: [3], 4: [4], 5: [5], 6: [6], 7: list(range(7, 36))},
    2: {0: [0], 1: [1], 2: [2], 3: [3], 4: [4], 5: [5], 6: [6], 7: [7, 8, 10, 27, 20, 32], 8: [9, 22], 9: [11, 33],
        10: [12, 28], 11: [13, 21], 12: [14, 24], 13: [15, 18], 14: [16, 23], 15: [17], 16: [19],
        255: [25, 26, 29, 30, 31, 34, 35]},
    3: dict([(i, [i]) for i in range(25)] + [(255, list(range(25, 36)))]),
}
