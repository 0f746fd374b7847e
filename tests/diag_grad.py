"""Diagnostic: HIP vs f32 oracle vs f64 oracle gradients (run on the GPU box)."""
import sys, os, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[R, R+'/dqo-map_amd', R+'/tests']
from dqo_harness import scenes
from oracle import oracle_lib as ol
import util_rast as U
P=int(sys.argv[1]) if len(sys.argv)>1 else 3000
cam, sc = scenes.make_config(1, P=P)
rng=np.random.default_rng(0)
dL=(rng.normal(size=(3,cam.H,cam.W)).astype(np.float32), rng.normal(size=(1,cam.H,cam.W)).astype(np.float32))
h,hg=U.run_hip(cam,sc,dL=dL)
o,r,og=U.run_oracle(ol,cam,sc,dL=dL)
o64=ol.OracleRasterizer(np.float64)
st=U.oracle_settings(ol,cam)
r64=o64.forward(st, sc["xyz"], sc["opacity"], cam.world_view_transform, cam.full_proj_transform, cam.camera_center, shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
g64=o64.backward(dL[0],dL[1])
og64=dict(means3D=g64.means3D,opacity=g64.opacity,scales=g64.scales,rotations=g64.rotations,sh=g64.sh)
bad=(h['hit_depth']!=r['hit_depth'])|(h['hit_color']!=r['hit_color'])
bad64=(r64.hit_depth!=r['hit_depth'])|(r64.hit_color!=r['hit_color'])
print('mismatch px hip-vs-o32', bad.sum(), ' o64-vs-o32', bad64.sum(), 'color', np.abs(h['color']-r['color']).max(), 'depth', np.abs(h['depth']-r['depth']).max())
for k in og:
    a=hg[k].reshape(og[k].shape); b=og[k]; c=og64[k].reshape(og[k].shape)
    s=np.abs(c).max()
    e=np.abs(a-b); i=np.unravel_index(e.argmax(), e.shape)
    print(f"{k:10s} hip-o32 {e.max()/s:.2e} at {i} hip {a[i]:.6g} o32 {b[i]:.6g} o64 {c[i]:.6g} | hip-o64 {np.abs(a-c).max()/s:.2e} o32-o64 {np.abs(b-c).max()/s:.2e} maxmag {s:.4g}")
