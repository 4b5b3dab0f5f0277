"""Checkpoint / metrics-file / early-stopping compatibility with the reference's training script.

* `save_checkpoint` writes exactly the dict of training_multitask.py:243-249 (keys 'epoch', 'model_state_dict',
  'optimizer_state_dict', 'scheduler', 'val_loss'); model keys are the reference's, the optimizer state is in
  torch.optim.Adam's layout (optim.FusedAdam.state_dict), so files interchange with the reference in both directions.
* `load_pretrained_model` is src/utils/models.py:19-36 (same name, same ValueError on a missing file).
* `write_metrics_file` (src/utils/miscellany.py:155-169), `METRICS_HEADER` (training_multitask.py:216-217) and
  `metrics_row` (:271-275, including the stray blank before Validation_dice that the reference's f-string emits).
* `EarlyStopping` is the patience logic of :238-251, :277-280.
"""
from __future__ import annotations

import logging
import os
from typing import Optional

import torch

METRICS_HEADER = "epoch,LR,Train_loss,Validation_loss,Train_dice,Validation_dice,Train_acc,Train_F1,Validation_acc,Validation_F1"   # training_multitask.py:216-217


def save_checkpoint(path: str, epoch: int, model, optimizer, val_loss: float) -> None:
    torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                "scheduler": "scheduler", "val_loss": val_loss}, path)


def load_pretrained_model(model, ckpt_path: str, optimizer=None):
    if os.path.isfile(ckpt_path):
        checkpoint = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        model.load_state_dict(checkpoint["model_state_dict"])
        if optimizer is not None and checkpoint.get("optimizer_state_dict") is not None:
            optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
        logging.info(f"Loaded checkpoint '{ckpt_path}'. Last epoch: {checkpoint['epoch']}")
    else:
        raise ValueError(f"\n\t-> No checkpoint found at '{ckpt_path}'")
    return model


def metrics_row(epoch: int, lr: float, train_loss: float, val_loss: float, train_dice: float, val_dice: float,
                train_acc: float, train_f1: float, val_acc: float, val_f1: float) -> str:
    return (f"{epoch},{lr:.8f},{train_loss:.4f},{val_loss:.4f},{train_dice:.4f}, {val_dice:.4f},{train_acc:.4f},"
            f"{train_f1:.4f},{val_acc:.4f},{val_f1:.4f}")


def write_metrics_file(path_file: str, text_to_write: str, close: bool = True) -> None:
    """src/utils/miscellany.py:155-169: append one line (the caller writes METRICS_HEADER first, as :216-217 does)."""
    with open(path_file, "a") as f:
        f.write(text_to_write)
        f.write("\n")


class EarlyStopping:
    """best-validation tracking of training_multitask.py:238-251 and the stop rule of :277-280."""

    def __init__(self, max_patience: int):
        self.max_patience = int(max_patience)
        self.best = float("inf")
        self.patience = 0

    def update(self, val_loss: float) -> bool:
        """True when this epoch is the new best (the caller saves the checkpoint)."""
        if val_loss < self.best:
            self.patience = 0
            self.best = val_loss
            return True
        self.patience += 1
        return False

    @property
    def should_stop(self) -> bool:
        return self.patience > self.max_patience
