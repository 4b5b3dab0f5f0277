import os, sys, torch
sys.path.insert(0, os.getcwd())
from multi_task_breast_cancer_amd.miscellany import seed_everything
from multi_task_breast_cancer_amd.nets import MTUNetPlusPlus
from multi_task_breast_cancer_amd.optim import FusedAdam
from multi_task_breast_cancer_amd.synthetic import synthetic_batch
from multi_task_breast_cancer_amd.trainer import FusedTrainStep
from oracle import torch_oracle as O
import copy
dev = torch.device("cuda:0"); torch.set_num_threads(16)
seed_everything(1993)
prod = MTUNetPlusPlus(in_channels=1, out_channels=1, n_classes=3, deep_supervision=True)
ref = O.build_oracle_model("MTUNetPlusPlus", 1, 1, 3, True); ref.load_state_dict(prod.state_dict())
ref64 = copy.deepcopy(ref).double()
prod = prod.to(dev)
step = FusedTrainStep(prod, FusedAdam(prod, lr=5e-4, eps=1e-4), alpha=0.35)
ropt = torch.optim.Adam(ref.parameters(), lr=5e-4, eps=1e-4)
ropt64 = torch.optim.Adam(ref64.parameters(), lr=5e-4, eps=1e-4)
for s in range(60):
    img, mask, label = synthetic_batch(8, 64, 64, seed=s, device=torch.device("cpu"))
    lh = step(img.to(dev), mask.to(dev), label.to(dev))
    lo = O.train_step(ref, ropt, img, mask, label, 0.35, True, 3)
    l64 = O.train_step(ref64, ropt64, img.double(), mask.double(), label, 0.35, True, 3)
    if s < 12 or s % 8 == 0:
        pd = max((a.detach().cpu().double() - b.detach()).abs().max().item() for a, b in zip(prod.parameters(), ref64.parameters()))
        od = max((a.detach().double() - b.detach()).abs().max().item() for a, b in zip(ref.parameters(), ref64.parameters()))
        print(f"step {s+1:3d} loss hip {float(lh[0]):.6f} oracle32 {float(lo[0]):.6f} oracle64 {float(l64[0]):.6f} | max|dparam| hip-vs-64 {pd:.2e} oracle32-vs-64 {od:.2e}", flush=True)
