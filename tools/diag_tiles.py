"""Diagnostic (GPU box): distribution of per-tile list lengths n, backward walk lengths L, per-pixel contributors."""
import sys, os, numpy as np, torch
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[R, R+'/dqo-map_amd']
from dqo_harness import scenes, mapping
import diff_gaussian_rasterization_depth as dgr
cfg=int(sys.argv[1]) if len(sys.argv)>1 else 3
P=int(sys.argv[2]) if len(sys.argv)>2 else None
cam, sc = scenes.make_config(cfg, P=P)
dev=torch.device('cuda')
params=mapping.GaussianParams(sc, dev)
st=mapping.make_settings(cam, dev)
class Ctx:
    def save_for_backward(self,*a): self.saved=a
    def mark_non_differentiable(self,*a): pass
c=Ctx()
a=params.activated()
with torch.no_grad():
    out=dgr._RasterizeGaussians.forward(c, a['xyz'], a['shs'], torch.Tensor([]), a['opacity'], a['scales'], a['rotations'], torch.Tensor([]), None, st)
img=c.saved[10]
W,H=cam.W,cam.H; T=((W+15)//16)*((H+15)//16); HW=W*H
al=lambda n:(n+255)//256*256
off=0
def take(nbytes, dt):
    global off
    t=img[off:off+nbytes].view(dt).cpu().numpy(); off+=al(nbytes); return t
tile_count=take(4*T*64, torch.int32)[::64]; tflag=take(4*T, torch.int32); cursor=take(4*T*64, torch.int32); ranges=take(8*T, torch.int32).reshape(T,2); walk4=take(16*T, torch.int32).reshape(T,4); order=take(4*T, torch.int32)
final_T=take(4*HW, torch.float32); ncon=take(4*HW, torch.int32); hitpos=take(4*HW, torch.int32) & 0x7fffffff
n=ranges[:,1]-ranges[:,0]
q=lambda x:[int(np.quantile(x,p)) for p in (0.5,0.9,0.99,1.0)]
print('candidates',c.num_rendered,'instances',int(n.sum()),'tiles',T,'n: mean',n.mean(),'q50/90/99/max',q(n))
for lim in (64,128,256,512,1024,2048): print('tiles with n >',lim,':',int((n>lim).sum()),' instances in them', int(n[n>lim].sum()))
walk=walk4.max(1)
print('walk L per quadrant: mean',walk4.mean(),'q',q(walk4.ravel()),' sum(L4)/(4 sum n)=',walk4.sum()/(4*n.sum()))
print('n_contrib per pixel: mean',ncon.mean(),'q',q(ncon),' hit_pos mean',hitpos.mean(),'q',q(hitpos), 'frac no-hit', (hitpos==0).mean())
# live (quadrant, instance) pairs
binning=c.saved[9]; cap=c.inst_capacity
offb=al(8*cap)+al(4*cap)+al(4*cap)+al(4*cap)
lq=binning[offb:offb+4*cap].cpu().numpy().reshape(4,cap)
live=0
for t in range(T):
    for qd in range(4):
        L=walk4[t,qd]
        if L: live+=int(lq[qd, ranges[t,0]:ranges[t,0]+L].sum())
print('live (quadrant, instance) pairs walked by the backward:', live, ' per instance', live/max(1,n.sum()))
vis=(out[8]>0).sum().item(); print('visible',vis)
