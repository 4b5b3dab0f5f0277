import sys, torch, torch.nn.functional as F
sys.path.insert(0,'.')
from multi_task_breast_cancer_amd import ops
DEV='cuda:0'
g=torch.Generator().manual_seed(0)
N,C,H,W=2,512,4,4
z=torch.randn(N,C,H,W,generator=g)
gamma=torch.ones(C); beta=torch.zeros(C)
dyc=torch.randn(N,C,1,1,generator=g).expand(N,C,H,W).contiguous()/16
for name,dy in (('const',dyc),('rand',torch.randn(N,C,H,W,generator=g))):
    zr=z.double().requires_grad_(True); gr=gamma.double().requires_grad_(True); br=beta.double().requires_grad_(True)
    y=F.leaky_relu(F.instance_norm(zr,weight=gr,bias=br,eps=1e-5),0.1); y.backward(dy.double())
    z32=z.clone().requires_grad_(True); g32=gamma.clone().requires_grad_(True); b32=beta.clone().requires_grad_(True)
    y32=F.leaky_relu(F.instance_norm(z32,weight=g32,bias=b32,eps=1e-5),0.1); y32.backward(dy)
    yg,mean,rstd=ops.instnorm_lrelu_fwd(z.to(DEV),gamma.to(DEV),beta.to(DEV),1e-5,0.1)
    dz,dg,db=ops.instnorm_lrelu_bwd(z.to(DEV),dy.to(DEV),mean,rstd,gamma.to(DEV),beta.to(DEV),1e-5,0.1)
    rel=lambda a,b:((a.double().cpu()-b).norm()/b.norm()).item()
    print(name,'fwd',rel(yg,y.detach()),'dz ours',rel(dz,zr.grad),'t32',rel(z32.grad,zr.grad),'dgamma',rel(dg,gr.grad),rel(g32.grad,gr.grad),'dbeta',rel(db,br.grad),rel(b32.grad,br.grad))
# direct conv wgrad/dgrad at 4x4 512->512
x=torch.randn(2,512,4,4,generator=g); w=torch.randn(512,512,3,3,generator=g)*0.02; dzz=torch.randn(2,512,4,4,generator=g)
xr=x.double().requires_grad_(True); wr=w.double().requires_grad_(True)
F.conv2d(xr,wr,None,padding=1).backward(dzz.double())
dw,_=ops.conv3x3_wgrad([x.to(DEV)],dzz.to(DEV),tuple(w.shape))
dx=torch.zeros_like(x).to(DEV); ops.conv3x3_dgrad(dzz.to(DEV),w.to(DEV),[dx],[0],packed=None)
print('direct wgrad',rel(dw,wr.grad),'dgrad',rel(dx,xr.grad))
