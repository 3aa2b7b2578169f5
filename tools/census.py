"""SIMT execution census of the skip kernel (where do the wave-instructions go?)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vokselis_amd as V
W, H = 1920, 1080
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
for name, gen in (("standin", lambda: V.VolumeTexture.generate_standin(ctx, (256,) * 3, layout=V.LAYOUT_PACKED_PAIRS)),
                  ("fog", lambda: V.VolumeTexture.generate_fog(ctx, (256,) * 3, layout=V.LAYOUT_PACKED_PAIRS))):
    gen(); ctx.update()
    for fl, nm in ((0, "skip"), (V.RENDER_NO_SKIP, "noskip")):
        ctx.reset_step_counts()
        V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=0.5, flags=fl | V.RENDER_COUNT).record(ctx)
        s_ref, s_samp = ctx.step_counts(); c = ctx.simt_census()
        print(json.dumps({"case": f"{name}_{nm}", "S_ref": s_ref, "S_sampled": s_samp, **c,
                          "lanes_per_loop_iter": c["lane_loop_iters"] / max(c["wave_loop_iters"], 1),
                          "lanes_per_sample": s_samp / max(c["wave_sample_execs"], 1),
                          "lanes_per_skip_iter": (s_ref - s_samp) / max(c["wave_skip_iters"], 1)}))
ctx.close()
