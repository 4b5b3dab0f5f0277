import sys, torch
sys.path.insert(0, '.')
from multi_task_breast_cancer_amd.miscellany import seed_everything
from multi_task_breast_cancer_amd.nets import MTnnUNet, MTUNetPlusPlus
from oracle import torch_oracle as O
DEV = torch.device('cuda:0')
arch, N, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
seed_everything(11)
prod = MTnnUNet(1,1,3) if arch == 'MTnnUNet' else MTUNetPlusPlus(in_channels=1,out_channels=1,n_classes=3,deep_supervision=True)
ref = O.build_oracle_model(arch,1,1,3,True); ref.load_state_dict(prod.state_dict()); ref = ref.double(); prod = prod.to(DEV)
img, mask, label = O.synthetic_batch(N,size,size,seed=size+N)
st = prod.compiled(N,size,size)
st.x.data.copy_(img.to(DEV)); st.programs['pack'].run(); st.programs['fwd'].run(); torch.cuda.synchronize()
outs = {}
def mk(name):
    def hook(m, i, o): outs.setdefault(name, []).append(o.detach())
    return hook
for name, mod in ref.named_modules():
    if name in st.plan.acts: mod.register_forward_hook(mk(name))
with torch.no_grad(): ref(img.double())
for name, act in st.plan.acts.items():
    if name not in outs: continue
    for k, want in enumerate(outs[name]):
        if tuple(want.shape) != tuple(act.data.shape): continue
        got = act.data.cpu().double()
        flips = ((got > 0) != (want > 0))
        nf = int(flips.sum())
        print('%-40s use%d relL2 %.3e flips %d %s' % (name, k, ((got-want).norm()/want.norm()).item(), nf, (want[flips].abs().tolist()[:4] if nf else '')))
