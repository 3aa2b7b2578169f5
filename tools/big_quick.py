"""C4 / C5 in a few seconds: single-frame launch time (HIP events, best of three groups) of the staged march on the config's own camera, the
fog and its dense-core variant, with a checksum of the rgba32f frame -- the A/B probe for staged-kernel experiments (tools/ab.py runs it
against library variants built by tools/variant.py).  usage: tools/big_quick.py [c5|c4] [--reps N]"""
import sys, os, json, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import variant
variant.use_variant_from_env()
import torch  # noqa: F401
import vokselis_amd as V

which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "c5"
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
n, fmt, W, H, seed = {"c5": (2048, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005), "c4": (1024, V.FMT_R16_FLOAT, 1920, 1080, 0x5EED0004)}[which]


def t(ctx, fn, iters, groups=3):
    for _ in range(2): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best


cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
out = {"lib": os.environ.get("VK_LIB", "product"), "config": which}
for core in (False, True):
    name = "core" if core else "fog"
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed, dense_core=core); ctx.update()
    p = V.RaycastPipeline(dt_scale=0.5)
    p.record(ctx)
    out[name + "_crc"] = "%08x" % zlib.crc32(ctx.read_backbuffer().tobytes())
    ctx.resize_backbuffer(W, H, V.OUT_RGBA16F)
    for _ in range(reps):
        out.setdefault(name + "_ms", []).append(round(t(ctx, lambda: p.record(ctx), 6 if which == "c5" else 20), 4))
    ctx.close()
print(json.dumps(out), flush=True)
