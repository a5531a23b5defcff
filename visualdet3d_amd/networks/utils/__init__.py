from .registry import (AUGMENTATION_DICT, BACKBONE_DICT, DATASET_DICT, DETECTOR_DICT, PIPELINE_DICT,  # noqa: F401
                       SAMPLER_DICT, Registry)
