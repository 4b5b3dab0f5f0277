"""The optimisation step of src/training_multitask.py:87-103 as ONE stream-ordered program, plus the
data-parallel layer the reference does not have: one process per GPU, equal contiguous shards of a
single seeded global permutation, bucketed gradient all-reduce (RCCL over xGMI) overlapped with backward,
1/world folded into the fused Adam.

    zero_grad -> fwd -> Dice(4 heads, 1/(j+1)) + Focal -> alpha-mix -> bwd -> [all-reduce] -> Adam

No host synchronisation happens inside a step: the loss scalars and the NaN flag stay on the device
(`FusedTrainStep.losses`), to be read every k steps (the reference syncs 3+N times per step).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import os

import numpy as np
import torch

from . import _lib as L
from . import switches as _sw


# ------------------------------------------------------------------------------------------------
# pure host logic (CPU-testable with gloo)
# ------------------------------------------------------------------------------------------------
@dataclass
class Bucket:
    start: int          # element range [start, end) of the flat gradient buffer
    end: int
    ready_op: int       # backward ops [0, ready_op] must have been issued before this bucket is reduced


def plan_buckets(slots: Sequence[Tuple[int, int, int]], flat_numel: int, n_buckets: int) -> List[Bucket]:
    """slots: (offset, numel, ready_at) per parameter in layout order.  Splits the flat buffer into
    `n_buckets` contiguous ranges of roughly equal size on parameter boundaries and returns them from the END of the
    buffer to its start: parameters are laid out in construction (= forward) order, so that is the order in which the
    backward pass finishes them, and -- unlike a sort by op index -- it is the same on every rank even when ranks
    build different step programs (the short last batch of an epoch): collectives must pair up.  A bucket whose
    `ready_op` lies behind its successor's only waits a little longer.  Ranges tile [0, flat_numel) exactly."""
    if not slots:
        return []
    n_buckets = max(1, min(n_buckets, len(slots)))
    cuts = [0]
    for i, (off, numel, _) in enumerate(slots):
        nxt = slots[i + 1][0] if i + 1 < len(slots) else flat_numel
        left = n_buckets - len(cuts)            # cuts still allowed after this bucket closes
        if left <= 0:
            break
        target = cuts[-1] + (flat_numel - cuts[-1]) / (left + 1)
        if nxt >= target and nxt < flat_numel:
            cuts.append(nxt)
    cuts.append(flat_numel)
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b <= a:
            continue
        ready = max((r for off, numel, r in slots if a <= off < b), default=0)
        out.append(Bucket(a, b, ready))
    out.sort(key=lambda bk: -bk.start)
    return out


def shard_positions(positions: np.ndarray, rank: int, world: int, global_batch: int, step: int) -> np.ndarray:
    """Batch `step` of a global index list, cut into `world` equal contiguous shards: rank r takes
    rows [r*G/world, (r+1)*G/world) of the global batch, so the union over ranks IS the single-GPU batch."""
    if global_batch % world:
        raise ValueError("global batch must divide evenly across ranks")
    per = global_batch // world
    lo = step * global_batch + rank * per
    return positions[lo:lo + per]


def global_permutation(n: int, seed: int, epoch: int) -> np.ndarray:
    """One seeded permutation of the (oversampled) index list per epoch, identical on every rank."""
    return np.random.Generator(np.random.PCG64(seed * 1_000_003 + epoch)).permutation(n)


def allreduce_buckets(flat_g: torch.Tensor, buckets: Sequence[Bucket], group=None, async_op: bool = False):
    """Sum-all-reduce every bucket of the flat gradient buffer (averaging = grad_scale 1/world in Adam)."""
    import torch.distributed as dist
    works = []
    for b in buckets:
        w = dist.all_reduce(flat_g[b.start:b.end], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


# ------------------------------------------------------------------------------------------------
# the fused step (HIP)
# ------------------------------------------------------------------------------------------------
class FusedTrainStep:
    """One optimisation step of training_multitask.py:87-103 as ONE stream-ordered program.  `cls_criterion`: "Focal" (config.yaml's
    default, FocalLoss alpha 1 gamma 2), "CE" (CrossEntropyLoss = gamma 0) -- experiment_init.py:232-262; with n_classes == 2 the model has
    ONE logit and the reference trains it with BCEWithLogitsLoss on the {0, 1} label (`:241-242`, training_multitask.py:83-84 leaves the
    label (N, 1)): the same program with the focal kernel's one-logit form."""

    def __init__(self, model, optimizer, alpha: float, inversely_weighted: bool = True, n_classes: int = 3,
                 distributed: bool = False, n_buckets: int = 4, focal_weight: Optional[torch.Tensor] = None,
                 cls_criterion: str = "Focal", graph: Optional[bool] = None):
        # graph: replay each compiled step as ONE hipGraph from its third call on (None: the MTBC_GRAPH switch).  The step is a static list of ~380
        # launches with every pointer resolved at plan time -- exactly what a graph holds; what changes from step to step (the batch, the learning
        # rate, Adam's bias corrections, the shard weight) lives in device buffers written BEFORE the replay.  Not under data parallel (the bucket
        # all-reduces are issued between program ranges by the host).
        self.graph = _sw.flag("MTBC_GRAPH") if graph is None else bool(graph)
        self._graphs = {}
        self.binary = n_classes == 2
        if self.binary != (getattr(model, "n_classes", n_classes) == 1):
            raise ValueError("n_classes does not match the model's classification head (n_classes == 2 <=> ONE logit)")
        if cls_criterion not in ("Focal", "CE"):
            raise ValueError(f"unknown classification criterion {cls_criterion!r} (Focal | CE; the binary head always trains with BCEWithLogits)")
        self.cls_gamma = 2.0 if cls_criterion == "Focal" else 0.0
        if cls_criterion == "CE" and focal_weight is not None:
            # torch.nn.CrossEntropyLoss(weight=w) divides by the sum of the samples' weights; FocalLoss (criterions.py:14-24) by N
            raise NotImplementedError("class-weighted CrossEntropyLoss normalises by the weights' sum: use the drop-in loop for it")
        if self.binary:
            focal_weight = None
        if n_classes > 3:
            raise NotImplementedError("the fused step covers the reference's label set {0, 1, 2} (n_classes <= 3)")
        self.model, self.opt = model, optimizer
        self.alpha, self.iw, self.n_classes = float(alpha), bool(inversely_weighted), n_classes
        self.focal_weight = focal_weight
        self.distributed = distributed
        self.n_buckets = n_buckets
        self.world = 1
        self.comm_stream = None
        if distributed:
            import torch.distributed as dist
            self.world = dist.get_world_size()
            self.comm_stream = torch.cuda.Stream()
            optimizer.grad_scale = 1.0 / self.world
            # the bucket all-reduces run on their own stream under the backward pass: keep CUs free for RCCL's resident
            # kernels so that the cooperative InstanceNorm teams (which need every member resident) never queue behind
            # them.  A property of the step programs this trainer builds (mtbc_instnorm_args.coop_reserve_cus), not of
            # the process.
            if self.world > 1 or "MTBC_COOP_RESERVE_CUS" in os.environ:
                reserve = int(_sw.get("MTBC_COOP_RESERVE_CUS"))
                if getattr(model, "coop_reserve_cus", 0) != reserve:
                    model.coop_reserve_cus = reserve
        self._st = None
        self.losses: Optional[torch.Tensor] = None      # device: [total, seg, cls, nan_flag]

    def _compiled(self, N: int, H: int, W: int):
        st = self.model.compiled(N, H, W, fused_loss={"alpha": self.alpha, "inversely_weighted": self.iw, "focal_weight": self.focal_weight,
                                                      "binary": self.binary, "cls_gamma": self.cls_gamma})
        if st is not self._st:
            self._st = st
            self.model.grads_as_views()
        if st.buckets is None:          # readiness snapshot of THIS step program (model.slots is plan-time scratch)
            st.buckets = plan_buckets(st.slot_ready, self.model.flat_numel, self.n_buckets)
        return st

    def load_batch(self, image: torch.Tensor, mask: torch.Tensor, label: torch.Tensor, weight: Optional[float] = None):
        """H2D / D2D of training_multitask.py:82-84 into the plan's static buffers (one-hot on the device).
        `weight` = this rank's share n_local / n_batch of the global batch (EpochIndex.weights) when shards are NOT
        equal -- the short last batch of `DataLoader(drop_last=False)`, BUSI_dataloader.py:146: the local mean-loss
        gradient is then scaled by weight * world on the device, so that the summed, 1/world-averaged gradient is the
        global batch's."""
        N, _, H, W = image.shape
        st = self._compiled(N, H, W)
        st.x.data.copy_(image, non_blocking=True)
        st.mask.copy_(mask, non_blocking=True)
        lab = label.to(st.onehot.device, non_blocking=True).flatten()
        if self.binary:             # the (N, 1) float label itself is the target of the one-logit criterion
            st.onehot.copy_(lab.view(-1, 1).to(torch.float32))
        else:
            st.onehot.zero_()
            st.onehot.scatter_(1, lab.to(torch.int64).view(-1, 1), 1.0)
        st.grad_weight.fill_(1.0 if weight is None else float(weight) * self.world)
        return st

    def _reduce_all(self) -> None:
        allreduce_buckets(self.model.flat_g, self._st.buckets if self._st is not None and self._st.buckets else
                          plan_buckets([(0, self.model.flat_numel, 0)], self.model.flat_numel, 1))

    def _run_graph(self, st) -> torch.Tensor:
        """The step as a hipGraph replay: eager for the first two calls of a compiled step (lazily created buffers, kernel attributes), captured at the
        third, replayed afterwards.  A graph holds addresses: it is keyed by the compiled step and by the optimizer's buffers and dropped when they move."""
        opt = self.opt
        opt.grad_scale = (1.0 / self.world) / getattr(st, "loss_scale", 1.0)
        opt.advance_dynamic()                     # step count, lr, bias corrections -> 12 bytes of device memory, in stream order, outside the graph

        def body():
            P = st.programs
            P["pack"].run(); P["fwd"].run(); P["loss"].run(); P["bwd"].run()
            opt.launch_dynamic()

        # the entry lives ON the compiled step (a dropped step takes its graph along; no address or id() can be reused under a stale graph)
        ents = st.__dict__.setdefault("_graph_ents", {})
        key = (id(self), opt.graph_key())
        ent = ents.get(id(self))
        if ent is None or ent[0] != key:
            ents[id(self)] = ent = [key, 0, None]
        self._graphs[id(st)] = ent                # (introspection: tests, tools)
        if ent[2] is None and ent[1] >= 2:
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                body()
            cur.wait_stream(side)
            ent[2] = g
        if ent[2] is not None:
            ent[2].replay()
        else:
            ent[1] += 1
            body()
        self.losses = st.plan.loss_out
        self._coop_err = self.model.coop_error_word()
        return self.losses

    def run(self, st) -> torch.Tensor:
        """One optimisation step on the batch already resident in the plan's buffers."""
        if self.graph and not self.distributed:
            return self._run_graph(st)
        P = st.programs
        P["pack"].run()
        P["fwd"].run()
        P["loss"].run()
        if not self.distributed:
            P["bwd"].run()
        else:
            cur = torch.cuda.current_stream()
            done = 0
            for b in st.buckets:
                upto = min(P["bwd"].n, b.ready_op + 1)
                if upto > done:
                    P["bwd"].run(done, upto - done)
                    done = upto
                ev = torch.cuda.Event()
                ev.record(cur)
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(ev)
                    allreduce_buckets(self.model.flat_g, [b])
            if done < P["bwd"].n:
                P["bwd"].run(done, P["bwd"].n - done)
            cur.wait_stream(self.comm_stream)
        self.opt.grad_scale = (1.0 / self.world) / getattr(st, "loss_scale", 1.0)
        self.opt.step(grads_in_flat=True)
        self.losses = st.plan.loss_out
        self._coop_err = self.model.coop_error_word()      # ONE word per model: every compiled step's kernels set it
        return self.losses

    def run_empty(self) -> None:
        """This rank's shard of the (short, last) global batch is EMPTY: contribute a zero gradient to the same
        collectives the other ranks issue, then apply the same update.  Needs one earlier real step (bucket layout)."""
        if not self.distributed:
            return
        if self._st is None:
            raise L.MtbcError("run_empty() before any real step: the bucket layout is not known yet")
        self.model.flat_g.zero_()
        for b in self._st.buckets:
            allreduce_buckets(self.model.flat_g, [b])
        self.opt.grad_scale = (1.0 / self.world) / getattr(self._st, "loss_scale", 1.0)
        self.opt.step(grads_in_flat=True)

    def __call__(self, image, mask, label, weight: Optional[float] = None) -> torch.Tensor:
        return self.run(self.load_batch(image, mask, label, weight))

    def check_nan(self) -> None:
        """The reference's NaN guard (criterions.py:72-76) plus the device-side protocol check of the cooperative
        InstanceNorm kernels, one device->host read when the caller chooses.  NaN -> log + exit(1) like the reference;
        a cooperative-kernel failure (a team member was not resident: results are garbage) raises MtbcError."""
        if self.losses is None:
            return
        err = getattr(self, "_coop_err", None)
        vals = torch.cat([self.losses[3:4].double(), (err if err is not None else self.losses[3:4] * 0).double()]).cpu().tolist()
        if vals[1] != 0.0:
            raise L.MtbcError("cooperative InstanceNorm: a mailbox poll timed out (team members were not co-resident, e.g. "
                              "another stream or process held the CUs): activations and gradients of this and later steps "
                              "are invalid. Raise MTBC_COOP_RESERVE_CUS, or set MTBC_NO_COOP=1.")
        if vals[0] != 0.0:
            import logging
            import sys
            logging.info("NaN in model loss!!")
            sys.exit(1)


def dice_score_from_counts(counts: torch.Tensor) -> float:
    """metrics.py:255-267 from {tp, fp, fn} float64 counts (mtbc_dice_counts)."""
    tp, fp, fn = (float(v) for v in counts.tolist())
    if tp + fn == 0:
        return 1.0 if tp + fp == 0 else 0.0
    return 2 * tp / (2 * tp + fp + fn)


def dice_counts(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    import ctypes as C
    out = torch.empty(3, dtype=torch.float64, device=logits.device)
    x, t = logits.contiguous().float(), target.contiguous().float()
    L.check(L.load().mtbc_dice_counts(x.data_ptr(), t.data_ptr(), x.numel(), out.data_ptr(),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "dice_counts")
    return out


# ------------------------------------------------------------------------------------------------
# validation / inference epoch
# ------------------------------------------------------------------------------------------------
class FusedEvalStep:
    """Forward + losses + metrics of `validate_one_epoch` (training_multitask.py:119-159) with no host round trip per
    batch: the step program runs pack -> forward -> fused Dice/Focal, `mtbc_dice_counts` gives the batch Dice of
    `process_segmentation_predicted` (:66-71: sigmoid(last head) > .5 against the mask, `dice_score_from_tensor`), and
    the confusion matrix of `processes_classification_predicted` (:34-63) accumulates on the device: multi-class =
    argmax of softmax vs argmax of the one-hot label (:41-51); binary head (n_classes == 2, ONE logit) = sigmoid > .5
    vs the {0,1} label (:53-61), classification loss BCEWithLogits (experiment_init.py:242) = the focal kernel's one-logit form
    inside the same step program.  `cls_criterion` ("Focal" | "CE") is the criterion the run TRAINS with (the reference validates with
    the training criterion, and `scheduler.step(val_loss)` runs on that value, training_multitask.py:234-237): with the same
    (alpha, weighting, criterion) the evaluation step shares the training step's compiled plan at equal (N, H, W).  `result()` reads
    everything back once and returns the reference's 6-tuple
    (avg_val_loss, avg_val_dice, val_acc, val_f1, avg_seg_val_loss, avg_cls_val_loss)."""

    def __init__(self, model, alpha: float, inversely_weighted: bool = True, n_classes: int = 3,
                 focal_weight: Optional[torch.Tensor] = None, cls_criterion: str = "Focal"):
        self.model, self.alpha, self.iw, self.n_classes = model, float(alpha), bool(inversely_weighted), n_classes
        self.binary = n_classes == 2
        if cls_criterion not in ("Focal", "CE"):
            raise ValueError(f"unknown classification criterion {cls_criterion!r} (Focal | CE; the binary head always evaluates BCEWithLogits)")
        self.cls_gamma = 2.0 if cls_criterion == "Focal" else 0.0
        if cls_criterion == "CE" and focal_weight is not None:       # same refusal as FusedTrainStep
            raise NotImplementedError("class-weighted CrossEntropyLoss normalises by the weights' sum: use the drop-in loop for it")
        if n_classes > 3:       # the confusion matrix is the reference's 3 x 3 (f1_score(labels=[0, 1, 2]), training_multitask.py:155)
            raise NotImplementedError("FusedEvalStep covers the reference's label set {0, 1, 2} (n_classes <= 3)")
        if self.binary != (getattr(model, "n_classes", n_classes) == 1):
            raise ValueError("n_classes does not match the model's classification head")
        self.focal_weight = None if self.binary else focal_weight
        self._coop_err = None
        self.reset()

    def reset(self) -> None:
        self._acc = None          # device float64: [sum total, sum seg, sum cls, sum dice, batches]
        self._conf = None         # device int64 (3, 3): rows = ground truth, cols = prediction (f1 is asked for labels 0,1,2)

    @torch.no_grad()
    def __call__(self, image: torch.Tensor, mask: torch.Tensor, label: torch.Tensor) -> None:
        N, _, H, W = image.shape
        st = self.model.compiled(N, H, W, fused_loss={"alpha": self.alpha, "inversely_weighted": self.iw,
                                                      "focal_weight": self.focal_weight, "binary": self.binary, "cls_gamma": self.cls_gamma})
        st.x.data.copy_(image, non_blocking=True)
        st.mask.copy_(mask, non_blocking=True)
        dev = st.plan.loss_out.device
        lab = label.to(dev, non_blocking=True).flatten()
        if self.binary:
            st.onehot.copy_(lab.view(-1, 1).to(torch.float32))
        else:
            st.onehot.zero_()
            st.onehot.scatter_(1, lab.to(torch.int64).view(-1, 1), 1.0)
        P = st.programs
        P["pack"].run()
        P["fwd"].run()
        P["loss"].run()
        self._coop_err = self.model.coop_error_word()
        if self._acc is None:
            self._acc = torch.zeros(5, dtype=torch.float64, device=dev)
            self._conf = torch.zeros(3, 3, dtype=torch.int64, device=dev)
        counts = dice_counts(st.segs[-1].data, st.mask)                 # {tp, fp, fn} float64 on the device
        tp, fp, fn = counts[0], counts[1], counts[2]
        empty_gt = (tp + fn) == 0
        dice = torch.where(empty_gt, torch.where((tp + fp) == 0, torch.ones_like(tp), torch.zeros_like(tp)),
                           2 * tp / torch.clamp(2 * tp + fp + fn, min=1.0))          # metrics.py:255-267
        logit = st.logits.data.view(N, -1)
        self._acc[:3] += st.plan.loss_out[:3].double()
        if self.binary:
            pred = (torch.sigmoid(logit[:, 0]) > 0.5).to(torch.int64)
            gt = lab.to(torch.int64)
        else:
            pred = logit.argmax(dim=1)
            gt = st.onehot.argmax(dim=1)
        self._acc[3] += dice
        self._acc[4] += 1
        self._conf.view(-1).index_add_(0, gt * 3 + pred, torch.ones_like(gt))

    def result(self):
        if self._acc is None:
            raise L.MtbcError("FusedEvalStep.result() before any batch was evaluated")
        err = self._coop_err
        if err is not None and int(err.item()) != 0:
            raise L.MtbcError("cooperative InstanceNorm: a mailbox poll timed out (team members were not co-resident): the "
                              "activations of this evaluation are invalid")
        acc = self._acc.cpu().tolist()
        conf = self._conf.cpu().numpy().astype(np.float64)
        nb = max(acc[4], 1.0)
        total = conf.sum()
        accuracy = float(np.trace(conf) / total) if total else 0.0
        # sklearn f1_score(labels=[0,1,2], average='weighted') (:155): per-class F1 weighted by support
        support = conf.sum(axis=1)
        tp = np.diag(conf)
        denom = conf.sum(axis=0) + support
        f1c = np.divide(2 * tp, denom, out=np.zeros_like(tp), where=denom > 0)
        f1w = float((f1c * support).sum() / support.sum()) if support.sum() else 0.0
        return acc[0] / nb, acc[3] / nb, accuracy, f1w, acc[1] / nb, acc[2] / nb


def validate_one_epoch(step: FusedEvalStep, loader, device) -> tuple:
    """training_multitask.py:119-159 on the fused evaluation step; `loader` yields the reference's batch dicts."""
    step.reset()
    for data in loader:
        step(data["image"].to(device), data["mask"].to(device), data["label"].to(device))
    return step.result()
