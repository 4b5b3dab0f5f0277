"""String -> object factory with the reference's names and signatures (src/utils/experiment_init.py:130-318),
restricted to the multi-task training path.  Unknown names raise ValueError (the reference silently builds an
empty nn.Module, experiment_init.py:160-163 -- stricter here)."""
from __future__ import annotations

import logging
import sys
from pathlib import Path

import torch
from torch.optim.lr_scheduler import CosineAnnealingLR, ReduceLROnPlateau

from .criterions import DiceLoss, FocalLoss
from .nets import MTnnUNet, MTUNetPlusPlus
from .optim import FusedAdam


def count_parameters(model: torch.nn.Module) -> int:
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def init_multitask_model(architecture: str, sequences: int = 1, regions: int = 1, n_classes: int = 2, width: int = 48,
                         save_folder: Path = None, deep_supervision: bool = False) -> torch.nn.Module:
    logging.info(f"Creating {architecture} model")
    logging.info(f"The model will be fed with {sequences} sequences")
    if architecture == "MTUNetPlusPlus":
        model = MTUNetPlusPlus(in_channels=sequences, out_channels=regions, n_classes=n_classes,
                               deep_supervision=deep_supervision)
    elif architecture == "MTnnUNet":
        model = MTnnUNet(sequences=sequences, regions=regions, n_classes=n_classes)
    else:
        raise ValueError(f"The model selected ({architecture!r}) is not on the MI355X hot path. Choose "
                         "'MTnnUNet' or 'MTUNetPlusPlus'.")
    if save_folder is not None:
        save_folder = Path(save_folder)
        save_folder.mkdir(parents=True, exist_ok=True)
        with (save_folder / "model.txt").open("w") as f:
            print(model, file=f)
    logging.info(f"Total number of trainable parameters: {count_parameters(model)}")
    return model


def init_optimizer(model: torch.nn.Module, optimizer: str, learning_rate: float = 0.001):
    """experiment_init.py:177-196.  'Adam' (config.yaml) is the fused HIP optimizer; 'SGD' / 'AdamW' / the reference's
    SGD fallback for unknown names are torch's own optimizers over the same parameters (drop-in loop only)."""
    if optimizer == "Adam":
        return FusedAdam(model, lr=learning_rate, eps=1e-4)          # experiment_init.py:187: eps=1e-4
    if optimizer == "SGD":
        return torch.optim.SGD(model.parameters(), lr=learning_rate, momentum=0.9, nesterov=True)
    if optimizer == "AdamW":
        return torch.optim.AdamW(model.parameters(), lr=learning_rate)
    logging.info(f"The optimizer '{optimizer}' is not recognized. SGD will be used instead.")
    return torch.optim.SGD(model.parameters(), lr=0.001, momentum=0.9, nesterov=True)


def init_criterion_segmentation(loss_function: str = "dice") -> torch.nn.Module:
    """experiment_init.py:199-232.  'DICE' (config.yaml) is the HIP kernel; 'BCE' is torch's module as in the
    reference; the remaining names are MONAI losses (not installed here, not on the hot path) -> same exit as the
    reference takes for an unknown name."""
    if loss_function == "DICE":
        return DiceLoss(include_background=True, sigmoid=True, smooth_dr=1, smooth_nr=1, squared_pred=True)
    if loss_function == "BCE":
        return torch.nn.BCEWithLogitsLoss()
    logging.info("Select a loss function allowed on the MI355X hot path: ['DICE', 'BCE']")
    sys.exit()


def init_criterion_classification(n_classes: int = 2, classes_weighted=None, classification_criterion="CE"):
    """experiment_init.py:235-263.  The 3-class Focal criterion (config.yaml:19) is the HIP kernel that the fused step
    also uses; binary (BCEWithLogitsLoss) and CrossEntropyLoss are torch's own modules, exactly as in the reference --
    they run on the HIP model's outputs through autograd (drop-in loop only; FusedTrainStep is Focal-only)."""
    if n_classes == 2:
        return torch.nn.BCEWithLogitsLoss()
    weight = None
    if classes_weighted:
        freq = torch.tensor(classes_weighted, dtype=torch.float)
        cw = 1.0 / freq
        weight = (cw / cw.sum()).to("cuda")
    if classification_criterion == "Focal":
        return FocalLoss(alpha=1, gamma=2, reduction="mean", weight=weight)
    return torch.nn.CrossEntropyLoss(reduction="mean", weight=weight)


def init_lr_scheduler(optimizer, scheduler: str = "cosine", t_max: int = 20, factor: float = 0.5,
                      min_lr: float = 1e-6, patience: int = 20):
    if scheduler == "plateau":
        return ReduceLROnPlateau(optimizer, mode="min", factor=factor, patience=patience, min_lr=min_lr)
    if scheduler == "cosine":
        return CosineAnnealingLR(optimizer, T_max=t_max, eta_min=min_lr)
    print("Select a scheduler allowed: ['plateau', 'cosine']")
    sys.exit()


def load_multitask_experiment_artefacts(config_data, config_model, config_opt, config_loss, n_augments, run_path):
    """experiment_init.py:301-318 -> (model, optimizer, segmentation_criterion, classification_criterion, scheduler)"""
    model = init_multitask_model(architecture=config_model["architecture"],
                                 sequences=config_model["sequences"] + n_augments,
                                 width=config_model["width"],
                                 n_classes=len(config_data["classes"]),
                                 deep_supervision=config_model["deep_supervision"],
                                 save_folder=Path(f"{run_path}/") if run_path is not None else None)
    # optional key, absent from the reference's config.yaml (default = the reference's fp32 arithmetic):
    #   model: {compute_dtype: bf16}   -> bf16 MFMA operands in the 3x3 convs, everything else fp32
    if config_model.get("compute_dtype") is not None and hasattr(model, "set_compute"):
        model.set_compute(str(config_model["compute_dtype"]))
    optimizer = init_optimizer(model=model, optimizer=config_opt["opt"], learning_rate=config_opt["lr"])
    segmentation_criterion = init_criterion_segmentation(loss_function=config_loss["function"])
    classification_criterion = init_criterion_classification(
        n_classes=len(config_data["classes"]), classes_weighted=config_data["classes_weighted"],
        classification_criterion=config_loss["classification_criterion"])
    scheduler = init_lr_scheduler(optimizer=optimizer, scheduler=config_opt["scheduler"],
                                  t_max=int(config_opt["t_max"]), patience=int(config_opt["patience"]),
                                  min_lr=float(config_opt["min_lr"]), factor=float(config_opt["decrease_factor"]))
    return model, optimizer, segmentation_criterion, classification_criterion, scheduler


def device_setup() -> str:
    """experiment_init.py:339-347"""
    if torch.cuda.is_available():
        logging.info("GPU will be used to train the model")
        return "cuda:0"
    logging.info("CPU will be used to train the model")
    return "cpu"
