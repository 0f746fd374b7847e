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
tile_count=take(4*T, torch.int32); tflag=take(4*T, torch.int32); cursor=take(4*T, torch.int32); ranges=take(8*T, torch.int32).reshape(T,2); walk=take(4*T, torch.int32); order=take(4*T, torch.int32)
final_T=take(4*HW, torch.float32); ncon=take(4*HW, torch.int32); hitpos=take(4*HW, torch.int32)
n=ranges[:,1]-ranges[:,0]
q=lambda x:[int(np.quantile(x,p)) for p in (0.5,0.9,0.99,1.0)]
print('N',c.num_rendered,'tiles',T,'n: mean',n.mean(),'q50/90/99/max',q(n))
print('walk L: mean',walk.mean(),'q',q(walk),' sum(L)/sum(n)=',walk.sum()/n.sum())
print('n_contrib per pixel: mean',ncon.mean(),'q',q(ncon),' hit_pos mean',hitpos.mean(),'q',q(hitpos), 'frac no-hit', (hitpos==0).mean())
# per-tile: fraction of pixels whose walk exceeds half the tile's L
need=np.maximum(ncon,hitpos).reshape(H,W)
gx=(W+15)//16
fr=[]
for t in np.argsort(-walk)[:10]:
    ty,tx=divmod(int(t),gx); blk=need[ty*16:(ty+1)*16, tx*16:(tx+1)*16]
    print('tile',t,'n',n[t],'L',walk[t],'px need q50/q90/max',q(blk.ravel())[:2], blk.max(), 'active px at L/2', int((blk>walk[t]/2).sum()))
# work estimates
print('sum over tiles of L (wave-entries bwd):', walk.sum(), ' max L', walk.max())
pix_need=need.astype(np.int64).sum()
print('sum over pixels of need:', pix_need, ' => avg lanes busy per wave-entry:', pix_need/ (walk.astype(np.int64).sum()*256+1)*256, 'of 256')
vis=(out[8]>0).sum().item(); print('visible',vis)
