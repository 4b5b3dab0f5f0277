import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multi_task_breast_cancer_amd.miscellany import seed_everything
from multi_task_breast_cancer_amd.nets import MTUNetPlusPlus
from multi_task_breast_cancer_amd.optim import FusedAdam
from multi_task_breast_cancer_amd.trainer import FusedTrainStep
from oracle import torch_oracle as O
DEV=torch.device('cuda:0')
N,S=int(sys.argv[1]),int(sys.argv[2])
img,mask,label=O.synthetic_batch(N,S,S,seed=21)
res={}
for mode in ('f32','bf16','f16'):
    seed_everything(1993)
    m=MTUNetPlusPlus(in_channels=1,out_channels=1,n_classes=3,deep_supervision=True).to(DEV); m.set_compute(mode)
    step=FusedTrainStep(m,FusedAdam(m,lr=1e-4,eps=1e-4),alpha=0.5)
    st=step.load_batch(img.to(DEV),mask.to(DEV),label.to(DEV)); l=step.run(st).cpu()
    res[mode]=(l,m.flat_g.clone(),[s.data.clone() for s in st.segs],st.logits.data.clone(),m)
l0,g0,s0,lg0,m0=res['f32']
for mode in ('bf16','f16'):
    l1,g1,s1,lg1,_=res[mode]
    print(mode,'loss',l0.tolist()[:3],l1.tolist()[:3],'seg rel',[((a-b).norm()/a.norm()).item() for a,b in zip(s0,s1)],'logits',lg0.flatten().tolist()[:3],lg1.flatten().tolist()[:3])
    rows=[]
    for name in m0._order:
        s=m0.slots[name]; a,b=g0[s.offset:s.offset+s.numel],g1[s.offset:s.offset+s.numel]
        if a.norm().item()/a.numel()**0.5>1e-8: rows.append((((a-b).norm()/a.norm()).item(),name))
    rows.sort(reverse=True)
    print('  worst:',[(round(r,4),n) for r,n in rows[:6]]); print('  median',rows[len(rows)//2][0])
