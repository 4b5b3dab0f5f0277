import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_task_breast_cancer_amd import ops
DEV='cuda:0'
cases=[(32,[24],24,256,256),(32,[24,24,24,24,48],24,256,256),(32,[96,96,96],96,64,64),(32,[48],48,128,128)]
for N,segs,co,H,W in cases:
    xs=[torch.randn(N,c,H,W,device=DEV) for c in segs]; cin=sum(segs)
    w=torch.randn(co,cin,3,3,device=DEV)*0.05
    pf,pd=ops.conv3x3_pack(w)
    for _ in range(3): z=ops.conv3x3_fwd(xs,w,None,packed=pf)
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): z=ops.conv3x3_fwd(xs,w,None,packed=pf)
    e.record(); e.synchronize()
    ms=s.elapsed_time(e)/10
    print(f"dbg={os.environ.get('MTBC_DBG','0')} {cin}->{co} @{H}: {ms:.3f} ms {2*N*H*W*cin*co*9/ms/1e9:.1f} TF")

print('--- wgrad')
for N,segs,co,H,W in cases:
    xs=[torch.randn(N,c,H,W,device=DEV) for c in segs]; cin=sum(segs)
    dz=torch.randn(N,co,H,W,device=DEV)
    for _ in range(2): ops.conv3x3_wgrad(xs,dz,(co,cin,3,3))
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): ops.conv3x3_wgrad(xs,dz,(co,cin,3,3))
    e.record(); e.synchronize()
    ms=s.elapsed_time(e)/5
    print(f"dbg={os.environ.get('MTBC_DBG','0')} wgrad {cin}->{co} @{H}: {ms:.3f} ms {2*N*H*W*cin*co*9/ms/1e9:.1f} TF")
