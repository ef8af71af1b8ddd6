"""Constants of the reference's `constants.py` that the train path needs (data tables)."""

# order == diffusers cross-attention call order (constants.py:1-4)
UNET_LAYERS = ["IN01", "IN02", "IN04", "IN05", "IN07", "IN08", "MID"] + [f"OUT{i:02d}" for i in range(3, 12)]

# DTU camera indexing (constants.py:13-31); 0-indexed while file names are 1-indexed
PATH_DTU_CALIBRATION_DIR = "data/dtu/Calibration/cal18"
DTU_TRAIN_IDX = [25, 22, 28, 40, 44, 48, 0, 8, 13]
DTU_EXCLUDE_IDX = [3, 4, 5, 6, 7, 16, 17, 18, 19, 20, 21, 36, 37, 38, 39]
DTU_TEST_IDX = [i for i in range(49) if i not in DTU_TRAIN_IDX + DTU_EXCLUDE_IDX]
DTU_SPLIT_IDXS = {"test": DTU_TEST_IDX, "train": DTU_TRAIN_IDX}

VALIDATION_PROMPTS = ["A photo of a {}", "A photo of a {} on a beach", "App icon of {}",
                      "A painting of {} in the style of Monet"]

# caption templates drawn by TextualInversionDataset.__getitem__ (constants.py:58-86): the 27
# "imagenet small" object templates
_T = ("a photo of a {}|a rendering of a {}|a cropped photo of the {}|the photo of a {}|a photo of a clean {}|"
      "a photo of a dirty {}|a dark photo of the {}|a photo of my {}|a photo of the cool {}|a close-up photo of a {}|"
      "a bright photo of the {}|a cropped photo of a {}|a photo of the {}|a good photo of the {}|a photo of one {}|"
      "a close-up photo of the {}|a rendition of the {}|a photo of the clean {}|a rendition of a {}|"
      "a photo of a nice {}|a good photo of a {}|a photo of the nice {}|a photo of the small {}|"
      "a photo of the weird {}|a photo of the large {}|a photo of a cool {}|a photo of a small {}")
IMAGENET_TEMPLATES_SMALL = _T.split("|")
assert len(IMAGENET_TEMPLATES_SMALL) == 27
