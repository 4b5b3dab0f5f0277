"""Per-op timing table of one training step (HIP events around every program op)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_task_breast_cancer_amd import _lib as L
from multi_task_breast_cancer_amd.experiment_init import init_multitask_model, init_optimizer
from multi_task_breast_cancer_amd.miscellany import seed_everything
from multi_task_breast_cancer_amd.synthetic import synthetic_batch
from multi_task_breast_cancer_amd.trainer import FusedTrainStep

arch = sys.argv[1] if len(sys.argv) > 1 else "MTUNetPlusPlus"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
DT = sys.argv[4] if len(sys.argv) > 4 else "f32"
dev = torch.device("cuda:0")
seed_everything(1993)
model = init_multitask_model(arch, 1, 1, 3, deep_supervision=True).to(dev)
model.set_compute(DT)
step = FusedTrainStep(model, init_optimizer(model, "Adam", 1e-4), alpha=0.5)
batch = synthetic_batch(B, S, S, 0, dev)
for _ in range(3):
    step(*batch)
torch.cuda.synchronize()
st = step._st
names = {v: k for k, v in vars(L).items() if k.startswith("OP_") and isinstance(v, int)}
rows = []
for pname in ("pack", "fwd", "loss", "bwd"):
    prog = st.programs[pname]
    for i in range(prog.n):
        op = prog.array[i]
        ms = []
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); prog.run(i, 1); e.record(); e.synchronize(); ms.append(s.elapsed_time(e))
        t = sorted(ms)[1]
        desc, fl = "", 0.0
        if op.kind in (L.OP_CONV3_FWD, L.OP_CONV3_DGRAD, L.OP_CONV3_WGRAD):
            a = op.u.conv3
            fl = 2.0 * a.N * a.H * a.W * a.Cin * a.Cout * 9
            desc = f"{a.Cin}->{a.Cout} @{a.H}x{a.W} segs{a.n_in} {'mfma' if (a.w_packed or (op.kind == L.OP_CONV3_WGRAD and a.Cin >= 8)) else 'direct'}"
            desc += f" {(a.N*a.H*a.W*4*(a.Cin+a.Cout))/1e6:.0f}MB"
        elif op.kind in (L.OP_CONVT_FWD, L.OP_CONVT_DGRAD, L.OP_CONVT_WGRAD):
            a = op.u.convT
            fl = 2.0 * a.N * a.H * a.W * a.Cin * a.Cout * a.k * a.k
            desc = f"{a.Cin}->{a.Cout} k{a.k} @{a.H}x{a.W}"
        elif op.kind in (L.OP_IN_FWD, L.OP_IN_BWD):
            a = op.u.inorm
            desc = f"C{a.C} @{a.H}x{a.W} bytes {a.N*a.C*a.H*a.W*4/1e6:.1f}MB"
        elif op.kind == L.OP_C8_PACK:
            a = op.u.c8pack
            desc = f"C{a.C} HW{a.HW} {a.N*a.C*a.HW*6/1e6:.0f}MB"
        rows.append((pname, i, names[op.kind], desc, t, fl))
tot = sum(r[4] for r in rows)
agg = {}
for r in rows:
    k = r[2]
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += r[4]; a[2] += r[5]
print(f"{arch} B={B} {S}x{S}: sum of per-op times {tot:.2f} ms")
for k, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:22s} n={n:3d} {ms:8.3f} ms {100*ms/tot:5.1f}%  {fl/ms/1e9 if fl else 0:7.1f} TF")
MINMS = float(os.environ.get("PEROP_MIN", "0.15"))
print(f"--- per op (>= {MINMS} ms)")
for r in rows:
    if r[4] >= MINMS:
        print(f"  {r[0]:4s} #{r[1]:3d} {r[2]:18s} {r[3]:44s} {r[4]:7.3f} ms {r[5]/r[4]/1e9 if r[5] else 0:7.1f} TF")
