"""multi_task_breast_cancer_amd -- MI355X (gfx950) native training path for the multi-task
segmentation + classification networks of caumente/multi_task_breast_cancer.

Host side (this package) mirrors the reference's config/model/criterion/optimizer surface
(src/utils/experiment_init.py:301-318); every tensor op runs in libmtbc_hip.so (include/mtbc.h).
"""
from ._lib import MtbcError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
