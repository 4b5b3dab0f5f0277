"""Config loading and seeding, mirroring src/utils/miscellany.py:17-30,78-96 (same names, same returns)."""
from __future__ import annotations

import logging
import os
import random
from pprint import pformat

import numpy as np
import torch
import yaml


def load_config_file(path: str):
    """Returns the five sections (model, optimizer, loss, training, data) of the reference's config.yaml."""
    with open(path) as cf:
        config = yaml.load(cf, Loader=yaml.FullLoader)
        logging.info(pformat(config))
    return config["model"], config["optimizer"], config["loss"], config["training"], config["data"]


def seed_everything(seed: int, cuda_benchmark: bool = False) -> None:
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = cuda_benchmark


def default_config() -> dict:
    """The reference's shipped src/config.yaml:1-53 as a dict (keys verbatim)."""
    return {
        "model": {"architecture": "MTnnUNet", "sequences": 1, "width": 24, "deep_supervision": True},
        "optimizer": {"opt": "Adam", "lr": 1e-4, "scheduler": "plateau", "patience": 20, "min_lr": 1e-6,
                      "decrease_factor": 0.5, "t_max": 40},
        "loss": {"function": "DICE", "inversely_weighted": True, "classification_criterion": "Focal"},
        "training": {"debug": False, "seed": 1993, "epochs": 200, "max_patience": 50, "CV": 4,
                     "cuda_benchmark": False, "alpha": 0.35, "threshold_postprocessing": 0,
                     "overlap_seg_based_on_class": True, "overlap_class_based_on_seg": True},
        "data": {"semantic_segmentation": False, "batch_size": 2, "train_size": 0.8,
                 "classes": ["benign", "malignant", "normal"], "classes_weighted": None, "oversampling": True,
                 "augmentation": {"CLAHE": False, "SOBEL": False, "brightness_brighter": False,
                                  "brightness_darker": False, "contrast_high": False, "contrast_low": False}},
    }
