import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_task_breast_cancer_amd import ops
DEV='cuda:0'
cases=[(32,[24],24,256,256),(32,[24,24,24,24,48],24,256,256),(32,[96,96,96],96,64,64),(32,[48],48,128,128),(32,[384],384,16,16),(32,[192],192,8,8),(3,[16],40,16,16),(5,[32],80,8,8),(1,[24,48],24,40,64)]
for N,segs,co,H,W in cases:
    g=torch.Generator().manual_seed(1)
    xs=[(torch.randn(N,c,H,W,generator=g)*1.0+0.3).to(DEV) for c in segs]; cin=sum(segs)
    dz=torch.randn(N,co,H,W,generator=g).to(DEV)
    ref,_=ops.conv3x3_wgrad(xs,dz,(co,cin,3,3),compute=0)
    for mode,name in ((0,'fp32'),(1,'bf16'),(2,'fp16')):
        for _ in range(2): out,_=ops.conv3x3_wgrad(xs,dz,(co,cin,3,3),compute=mode)
        torch.cuda.synchronize()
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): out,_=ops.conv3x3_wgrad(xs,dz,(co,cin,3,3),compute=mode)
        e.record(); e.synchronize(); ms=s.elapsed_time(e)/5
        rel=((out-ref).norm()/ref.norm()).item()
        print(f"wgrad {cin}->{co} @{H}x{W} N={N} {name}: {ms:.3f} ms {2*N*H*W*cin*co*9/ms/1e9:.1f} TF relL2 vs fp32 {rel:.2e}")

print('--- fwd / dgrad')
for N,segs,co,H,W in cases:
    g=torch.Generator().manual_seed(2)
    xs=[(torch.randn(N,c,H,W,generator=g)*1.0+0.3).to(DEV) for c in segs]; cin=sum(segs)
    w=(torch.randn(co,cin,3,3,generator=g)*(2.0/(9*cin))**0.5).to(DEV)
    dz=torch.randn(N,co,H,W,generator=g).to(DEV)
    pf,pd=ops.conv3x3_pack(w)
    ref=ops.conv3x3_fwd(xs,w,None,packed=pf)
    dref=[torch.zeros_like(x) for x in xs]; ops.conv3x3_dgrad(dz,w,dref,[0]*len(xs),packed=pd)
    for mode,name in ((0,'fp32'),(1,'bf16'),(2,'fp16')):
        if mode: lf,ld=ops.conv3x3_pack_lp(w,mode)
        else: lf,ld=pf,pd
        for _ in range(2): out=ops.conv3x3_fwd(xs,w,None,packed=lf,compute=mode)
        torch.cuda.synchronize()
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): out=ops.conv3x3_fwd(xs,w,None,packed=lf,compute=mode)
        e.record(); e.synchronize(); ms=s.elapsed_time(e)/5
        dx=[torch.zeros_like(x) for x in xs]
        s.record()
        for _ in range(5): ops.conv3x3_dgrad(dz,w,dx,[0]*len(xs),packed=ld,compute=mode)
        e.record(); e.synchronize(); ms2=s.elapsed_time(e)/5
        rel=((out-ref).norm()/ref.norm()).item()
        reld=max(((a-b).norm()/b.norm()).item() for a,b in zip(dx,dref))
        fl=2*N*H*W*cin*co*9
        print(f"conv {cin}->{co} @{H}x{W} N={N} {name}: fwd {ms:.3f} ms {fl/ms/1e9:.1f} TF rel {rel:.2e} | dgrad {ms2:.3f} ms {fl/ms2/1e9:.1f} TF rel {reld:.2e}")
