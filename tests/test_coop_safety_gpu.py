"""The cooperative (split-plane) InstanceNorm kernels need every member of a team resident at once.  These tests run
them beside a kernel that HOLDS compute units on another stream (tests/support/hog.hip: the stand-in for a resident RCCL
collective overlapping the backward pass) and check the failure path: the sticky device error word must reach the host.
"""
import ctypes as C
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from multi_task_breast_cancer_amd import _lib as L  # noqa: E402
from multi_task_breast_cancer_amd import engine  # noqa: E402
from multi_task_breast_cancer_amd.miscellany import seed_everything  # noqa: E402
from multi_task_breast_cancer_amd.nets import MTUNetPlusPlus  # noqa: E402
from multi_task_breast_cancer_amd.optim import FusedAdam  # noqa: E402
from multi_task_breast_cancer_amd.trainer import FusedEvalStep, FusedTrainStep  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

DEV = "cuda:0"
HOG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "support", "libhog.so")


def _hog():
    if not os.path.exists(HOG):
        pytest.fail(f"{HOG} is not built: run __graft_entry__.build()")
    lib = C.CDLL(HOG)
    lib.hog_launch.restype = C.c_int
    lib.hog_launch.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    return lib


def _model(dtype, reserve, no_coop=False):
    seed_everything(1993)
    m = MTUNetPlusPlus(in_channels=1, out_channels=1, n_classes=3, deep_supervision=True).to(DEV)
    m.set_compute(dtype)
    m.coop_reserve_cus = reserve
    return m


def _run_steps(m, batch, n_steps, after_warmup=None):
    step = FusedTrainStep(m, FusedAdam(m, lr=1e-4, eps=1e-4), alpha=0.5)
    step(*batch)                                       # builds the step programs (about a second of host work)
    torch.cuda.current_stream().synchronize()
    if after_warmup is not None:
        after_warmup()
    for _ in range(n_steps):
        losses = step(*batch)
    torch.cuda.current_stream().synchronize()          # this stream only: a device-wide sync would wait for the hog
    step.check_nan()                                   # raises on the cooperative kernels' error word
    st = step._st
    n_coop = sum(1 for prog in (st.programs["fwd"], st.programs["bwd"]) for i in range(prog.n)
                 if prog.array[i].kind in (L.OP_IN_FWD, L.OP_IN_BWD) and (prog.array[i].u.inorm.y8 or prog.array[i].u.inorm.dz8))
    return losses.cpu(), m.flat_p.detach().cpu().clone(), n_coop, step


@pytest.mark.parametrize("dtype,N,size", [("bf16", 2, 256), ("f16", 1, 512)])
def test_cooperative_step_beside_a_cu_hogging_kernel(dtype, N, size, monkeypatch):
    """64 CUs are held by another stream for the whole run; the step programs were planned with 64 CUs reserved
    (mtbc_instnorm_args.coop_reserve_cus, what a data-parallel FusedTrainStep sets): no poll may give up, and the
    results must be bit-identical to the same steps on an idle GPU -- and agree with the one-plane kernels."""
    hog = _hog()
    batch = tuple(t.to(DEV) for t in O.synthetic_batch(N, size, size, seed=11))
    l_idle, p_idle, n_coop, _ = _run_steps(_model(dtype, 64), batch, 2)
    assert n_coop > 0, "the configuration does not reach the cooperative kernels"

    side = torch.cuda.Stream()
    word = torch.zeros(4, dtype=torch.int32, device=DEV)
    t0 = [0.0]

    def launch_hog():
        rc = hog.hog_launch(64, 1500.0, C.c_void_p(word.data_ptr()), C.c_void_p(side.cuda_stream))
        assert rc == 0
        time.sleep(0.05)                               # the hog is running by now (1.5 s); the two steps take a fraction of that
        t0[0] = time.perf_counter()

    l_hog, p_hog, _, step = _run_steps(_model(dtype, 64), batch, 2, after_warmup=launch_hog)
    busy = time.perf_counter() - t0[0]
    assert busy < 1.0, f"the steps took {busy:.2f} s: they did not run beside the hog but behind it"
    assert not side.query(), f"the hog finished before the steps did ({busy:.2f} s): the test did not overlap them"
    side.synchronize()
    assert int(step._st.plan.coop_error_word().item()) == 0
    assert torch.equal(p_hog, p_idle) and torch.equal(l_hog, l_idle)

    # the same steps through the one-plane kernels + pack (statistics in another summation order: close, not equal)
    monkeypatch.setattr(engine, "_NO_COOP", True)
    l_ref, p_ref, n_coop_ref, _ = _run_steps(_model(dtype, 0), batch, 2)
    assert n_coop_ref == 0
    # (another arithmetic, not another order: this arm keeps the conv outputs in fp32 where the default plan stores them as fp16 -- the oracle check
    #  of this arm is test_16bit_mfma_modes_match_their_emulation[...-no_coop-40]; here: the two plans train the same function)
    assert abs(l_ref[0].item() - l_idle[0].item()) < 5e-3 * abs(l_ref[0].item())
    assert (p_ref - p_idle).abs().max().item() < 6.1e-4          # three Adam steps of lr 1e-4 move a weight by at most 3e-4 each way


def test_cooperative_error_word_reaches_the_host():
    """A poll that gives up sets a sticky word in the plan's state block; the training and evaluation steps must raise
    on it instead of feeding garbage to Adam (forced here by writing the word, as a timed-out kernel would)."""
    m = _model("bf16", 0)
    batch = tuple(t.to(DEV) for t in O.synthetic_batch(2, 256, 256, seed=3))
    step = FusedTrainStep(m, FusedAdam(m, lr=1e-4, eps=1e-4), alpha=0.5)
    step(*batch)
    step.check_nan()
    word = step._st.plan.coop_error_word()
    assert word is not None and int(word.item()) == 0
    word.fill_(1)
    step(*batch)                                       # the word is sticky: later launches do not clear it
    with pytest.raises(L.MtbcError, match="cooperative InstanceNorm"):
        step.check_nan()
    ev = FusedEvalStep(m, alpha=0.5)
    ev(*batch)
    with pytest.raises(L.MtbcError, match="cooperative InstanceNorm"):
        ev.result()
    word.zero_()
    ev.reset()
    ev(*batch)
    assert len(ev.result()) == 6


def test_error_word_of_one_step_program_is_seen_after_a_step_of_another():
    """ONE error word per model: a timeout in the plan of the short last batch (or of an evaluation batch of another
    size) must still raise when `check_nan()` runs after a later step on a DIFFERENT plan (the INTEGRATION.md recipe
    checks every 50 steps)."""
    m = _model("bf16", 0)
    step = FusedTrainStep(m, FusedAdam(m, lr=1e-4, eps=1e-4), alpha=0.5)
    full = tuple(t.to(DEV) for t in O.synthetic_batch(2, 256, 256, seed=3))
    short = tuple(t.to(DEV) for t in O.synthetic_batch(1, 256, 256, seed=4))
    step(*short)
    st_short = step._st
    step(*full)
    assert step._st is not st_short
    assert st_short.plan.coop_error_word().data_ptr() == step._st.plan.coop_error_word().data_ptr() == m.coop_error_word().data_ptr()
    step.check_nan()
    step(*short)
    st_short.plan.coop_error_word().fill_(1)           # what a timed-out kernel of the short-batch plan leaves
    step(*full)
    with pytest.raises(L.MtbcError, match="cooperative InstanceNorm"):
        step.check_nan()
    ev = FusedEvalStep(m, alpha=0.5)
    ev(*short)
    with pytest.raises(L.MtbcError, match="cooperative InstanceNorm"):
        ev.result()
    m.coop_error_word().zero_()


def test_reserve_is_a_property_of_the_step_program_not_of_the_process():
    """Two models in one process, one planned with 64 CUs reserved and one with none: neither changes the other's
    cooperative grids (the library keeps no process-wide setting), and both reproduce themselves bit for bit."""
    batch = tuple(t.to(DEV) for t in O.synthetic_batch(2, 256, 256, seed=5))
    la, pa, _, _ = _run_steps(_model("bf16", 64), batch, 1)
    lb, pb, _, _ = _run_steps(_model("bf16", 0), batch, 1)
    la2, pa2, _, _ = _run_steps(_model("bf16", 64), batch, 1)
    lb2, pb2, _, _ = _run_steps(_model("bf16", 0), batch, 1)
    assert torch.equal(pa, pa2) and torch.equal(pb, pb2) and torch.equal(la, la2) and torch.equal(lb, lb2)
