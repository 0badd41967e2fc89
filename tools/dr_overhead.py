import sys, time, os, ctypes as C
sys.path.insert(0,'/root/repo'); 
import numpy as np, torch
import cudaraytracer_amd as acr
from cudaraytracer_amd import distributed as D
ROOT='/root/repo'
data = acr.ImportData.load(os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box-cycles.json"))
view, keep = data.pack(); scene = acr.Scene(import_view=view)
core = acr.HipCore(0); core.upload_scene(scene.soa); core.set_camera(scene.camera)
dev=torch.device("cuda",0); stream=torch.cuda.Stream(dev); torch.cuda.set_stream(stream); core.set_stream(stream.cuda_stream)
W,H=(int(v) for v in (sys.argv[1] if len(sys.argv)>1 else '1920x1080').split('x'))
RB=int(sys.argv[2]) if len(sys.argv)>2 else 16
n=[0]
def fa():
    n[0]+=1
    return dict(seed=n[0], frames=0, max_bounces=1)
dr = D.DistributedRenderer(W,H,D.hip_render_tiles(core,W,H,W,16,fa),dev,tile_w=W,tile_h=16,rebalance_every=RB)
for _ in range(20): dr.render_frame()
dr.finish(); torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(300): dr.render_frame()
dr.finish(); torch.cuda.synchronize()
t=(time.perf_counter()-t0)/300*1e3
# host-only cost: how long the loop takes to ISSUE (no sync)
t0=time.perf_counter()
for _ in range(300): dr.render_frame()
ti=(time.perf_counter()-t0)/300*1e3
dr.finish(); torch.cuda.synchronize()
print(W,H,"rebalance",RB,"band path, 1 rank: ms/frame", round(t,4), " host issue ms/frame", round(ti,4))
