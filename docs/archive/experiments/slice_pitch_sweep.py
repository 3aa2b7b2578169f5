"""Staged march: the pitch between slices of the LDS window, padded to a residue of 256 bytes (64 banks x 4 B).
Lanes of a wave sit in different slices (rays of an 8x8 block are up to a slab apart along the major axis); with a
dense window the slice pitch is whatever rows x pieces x 16 comes to, and residues near 0 / 128 stack the slices on the
same banks.  ms per frame for C4 (f16) and C5 (u8), bonsai camera, per residue."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vokselis_amd as V

def run(n, fmt, W, H, seed, frames):
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed, layout=V.LAYOUT_STAGED)
    ctx.update()
    p = V.RaycastPipeline(dt_scale=0.5)
    def t():
        best = 1e9
        for _ in range(2): p.record(ctx)
        for _ in range(3):
            ctx.sync(); ctx.timer_begin()
            for _ in range(frames): p.record(ctx)
            ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / frames)
        return best
    out = {}
    for res in [None] + list(range(0, 256, 16)):
        ctx.set_param("stage_slice_res", 0 if res is None else res + 1)
        out["dense" if res is None else str(res)] = round(t(), 4)
    ctx.close()
    return out

which = sys.argv[1:] or ["c4", "c5"]
if "c4" in which: print(json.dumps({"c4_ms": run(1024, V.FMT_R16_FLOAT, 1920, 1080, 0x5EED0004, 6)}), flush=True)
if "c5" in which: print(json.dumps({"c5_ms": run(2048, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005, 3)}), flush=True)
