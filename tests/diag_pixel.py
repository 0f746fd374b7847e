import sys, os, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[R, R+'/dqo-map_amd', R+'/tests']
from dqo_harness import scenes
from oracle import oracle_lib as ol
import util_rast as U
cam, sc = scenes.make_config(1, P=10000)
h,_=U.run_hip(cam,sc)
o,r,_=U.run_oracle(ol,cam,sc)
d=np.abs(h['hit_color_weight']-r['hit_color_weight'])[0]
y,x=np.unravel_index(d.argmax(), d.shape)
print('pixel',x,y,'diff',d[y,x])
for k in ('color','depth','hit_color','hit_depth','hit_color_weight','hit_depth_weight','T_map'):
    print(k, h[k][:,y,x], r[k][:,y,x])
# walk the oracle list for that pixel
gx=(cam.W+15)//16; t=(y//16)*gx + x//16
rg=o.ctx('ranges')[t]; pl=o.ctx('point_list')[rg[0]:rg[1]]
m2=o.ctx('means2D'); co=o.ctx('conic_opacity')
T=np.float32(1.0)
for i,g in enumerate(pl):
    dx=m2[g,0]-np.float32(x); dy=m2[g,1]-np.float32(y)
    p=np.float32(-0.5)*(co[g,0]*dx*dx+co[g,2]*dy*dy)-co[g,1]*dx*dy
    if p>0: continue
    a=min(np.float32(0.99), co[g,3]*np.exp(p))
    if a<np.float32(1/255): 
        if a>np.float32(1/255)*0.999: print('  near-threshold', i,g,a, a-np.float32(1/255))
        continue
    print(i,g,'alpha',a,'T',T,'w',a*T, 'test_T', T*(1-a))
    T=T*(1-a)
    if T<1e-4: break
