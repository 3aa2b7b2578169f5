"""Build the HIP library a second time with extra -D switches, beside the product build, for A/B measurements on one box:

    python tools/variant.py NAME -DVK_EXP_SOMETHING=1 ...     ->  tools/_variants/NAME/libvokselis_hip.so

and run any tool against it with VK_LIB=tools/_variants/NAME/libvokselis_hip.so (tools/ab.py does both sides in child processes).
The product never reads VK_LIB: only tools that call use_variant_from_env() below do."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_variant(name, defines):
    import __graft_entry__ as g

    out = os.path.join(ROOT, "tools", "_variants", name)
    os.makedirs(out, exist_ok=True)
    tus, _ = g.hip_sources()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def one(tu):
        obj = os.path.join(out, os.path.splitext(os.path.basename(tu))[0] + ".o")
        r = subprocess.run([hipcc] + g.HIPCC_FLAGS + list(defines) + ["-c", "-o", obj, tu], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(one, tus))
    so = os.path.join(out, "libvokselis_hip.so")
    subprocess.run([hipcc] + g.HIP_LINK_FLAGS + ["-o", so] + objs, check=True)
    for o in objs:
        os.remove(o)
    return so


def use_variant_from_env():
    """Tools only: point the ctypes loader at $VK_LIB before the library is first loaded."""
    p = os.environ.get("VK_LIB")
    if p:
        from vokselis_amd import _native

        _native.LIB_PATH = os.path.join(ROOT, p) if not os.path.isabs(p) else p


if __name__ == "__main__":
    print(build_variant(sys.argv[1], sys.argv[2:]))
