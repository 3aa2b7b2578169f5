import sys, os, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import vokselis_amd as V
W,H=1920,1080
cam=V.Camera(3.0,-0.5,1.0,(0.0,0.0,0.0),W/H)
ctx=V.Context(W,H,cam,backbuffer=(W,H),out_format=V.OUT_RGBA32F)
ctx.set_camera_blob(cam.get_proj_view_matrix())
def t(p, iters=5):
    for _ in range(2): p.record(ctx)
    ctx.sync(); best=1e9
    for _ in range(3):
        ctx.timer_begin()
        for _ in range(iters): p.record(ctx)
        ctx.timer_end(); best=min(best, ctx.timer_elapsed_ms()/iters)
    return best
ps=V.RaycastPipeline(V.MODE_PROCEDURAL); pd=V.RaycastPipeline(V.MODE_PROCEDURAL, flags=V.RENDER_DEVICE_SINE)
ps.record(ctx); a=ctx.read_backbuffer().copy()
pd.record(ctx); b=ctx.read_backbuffer().copy()
clear=np.array([0.023,0.02,0.02,1.0],np.float32)
miss_a=(a==clear).all(-1); miss_b=(b==clear).all(-1)
d=np.abs(a-b)[...,:3]
hit=~miss_a
print(json.dumps({"ms_spec":t(ps),"ms_dev":t(pd,20),"miss_equal":bool((miss_a==miss_b).all()),"hit_share":float(hit.mean()),
 "mean_abs":float(d[hit].mean()),"max_abs":float(d.max()),"p99":float(np.percentile(d[hit],99)),
 "mean_spec":a[hit][:,:3].mean(0).tolist(),"mean_dev":b[hit][:,:3].mean(0).tolist(),
 "blur_corr": float(np.corrcoef(a[...,0].reshape(H//8,8,W//8,8).mean((1,3)).ravel(), b[...,0].reshape(H//8,8,W//8,8).mean((1,3)).ravel())[0,1])}))
ctx.close()
