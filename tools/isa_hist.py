#!/usr/bin/env python3
"""Static instruction mix of a kernel's loops, priced with the measured issue-cost table.

  python tools/isa_hist.py <kernel-name-regex> [--src vokselis_amd/csrc/vk_launch_cells.hip] [--blocks]

Compiles the translation unit to gfx950 assembly (hipcc -S, the product's own flags), finds the kernel whose mangled
name matches, and prints for every loop (the compiler's "Loop Header: Depth=n" annotations) the count of vector,
scalar, LDS and memory instructions and the issue cycles they cost a SIMD at 8 waves per SIMD by
profiles/r03_ubench_valu_issue_rate.txt (cycles per wave-instruction per SIMD, sustained clocks, s_memtime).
Used for the utilisation figures in profiles/r03_*_utilisation.txt: VALU utilisation of a kernel =
(dynamic instruction counts from the PMC pass x the loop's mean cost per instruction) / (SIMD cycles of the launch).
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# cycles per wave-instruction per SIMD at 8 waves per SIMD (profiles/r03_ubench_valu_issue_rate.txt)
COST = {
    "vop2_fast": 2.0,    # v_add/sub/mul/fmac_f32, v_add_u32, v_mov, v_and/xor (full rate)
    "fma": 2.0,          # v_fma_f32 (full rate; VOP3 encoding makes no difference)
    "half": 4.0,         # v_min/max_f32, v_fract/floor, every v_cvt_*, shifts, v_min_i32, v_cndmask_b32_e64
    "vop3_int": 4.0,     # v_lshl_add_u32 v_add3_u32 v_mad_i32_i24 v_med3 v_perm v_bfi
    "mix": 4.0,          # v_fma_mix_f32
    "pk": 4.0,           # v_pk_*_f32 (two results)
    "trans": 8.0,        # v_cos v_sin v_exp v_log v_rcp v_rsq v_sqrt
    "f64": 4.0,          # v_fma_f64 (measured); other f64 ops priced alike
    "cmp": 2.5,          # v_cmp_* (pair with an fma measured 4.7)
    "dpp": 4.0,
    "salu": 2.4,         # a scalar instruction beside a vector stream (v_fma + s_add pairs: 4.8 per pair); overlaps long vector ops
    "lds": 4.0, "vmem": 4.0, "smem": 2.4, "wait": 0.5, "branch": 2.4, "other": 3.0,
}


def classify(op: str) -> str:
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if not op.startswith("v_"):
        return "other"
    if "dpp" in op:
        return "dpp"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_fma_mix"):
        return "mix"
    if op.endswith("_f64") or "_f64_" in op:
        return "f64"
    if re.match(r"v_(cos|sin|exp|log|rcp|rsq|sqrt)_", op):
        return "trans"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "cmp"
    if re.match(r"v_(add|sub|subrev|mul)_f32(_e32)?$", op) or re.match(r"v_(add|sub|subrev)_u32(_e32)?$", op) or re.match(r"v_(fmac|fmamk|fmaak)_f32", op):
        return "vop2_fast"
    if re.match(r"v_(fma|mad)_f32", op) or re.match(r"v_(add|sub|mul)_f32_e64", op):
        return "fma"
    if re.match(r"v_(lshl_add|add3|mad_i32_i24|mad_u32_u24|med3|perm|alignbit|alignbyte|bfe|bfi|and_or|lshl_or|add_lshl|or3|xad|mad_u64|mul_lo|mul_hi|lshl_add_u64)", op):
        return "vop3_int"
    return "half"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--src", default=None, help="translation unit that instantiates the kernel (default: chosen from the kernel's name)")
    ap.add_argument("--asm", default=None, help="reuse this assembly file when newer than the sources")
    ap.add_argument("--blocks", action="store_true", help="print every basic block, not just the loops")
    ap.add_argument("--dump", action="store_true", help="print the instructions of the loops")
    a = ap.parse_args()
    if a.src is None:  # the march kernels are instantiated by three translation units (vokselis_amd/csrc/vk_ctx.hpp)
        tu = "vk_launch_staged.hip" if "staged" in a.kernel else ("vk_launch_compute.hip" if re.search("compute|procedural", a.kernel) else "vk_launch_cells.hip")
        a.src = os.path.join(ROOT, "vokselis_amd", "csrc", tu)
    if a.asm is None:
        a.asm = "/tmp/isa/%s.s" % os.path.splitext(os.path.basename(a.src))[0]
    deps = [a.src] + [os.path.join(os.path.dirname(a.src), f) for f in os.listdir(os.path.dirname(a.src)) if f.endswith(".hpp")]
    if not os.path.exists(a.asm) or any(os.path.getmtime(d) > os.path.getmtime(a.asm) for d in deps):
        os.makedirs(os.path.dirname(a.asm), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only", "-o", a.asm, a.src],
                       check=True, stderr=subprocess.DEVNULL)
    lines = open(a.asm).read().splitlines()
    starts = [i for i, ln in enumerate(lines) if re.match(r"^_Z\w+:", ln)]
    pick = [i for i in starts if re.search(a.kernel, lines[i])]
    if len(pick) != 1:
        sys.exit("kernel regex matches %d symbols:\n%s" % (len(pick), "\n".join(lines[i].split(":")[0] for i in pick[:20])))
    i0 = pick[0]
    i1 = next(i for i in range(i0, len(lines)) if re.match(r"^\.Lfunc_end\d+:", lines[i]))  # (a kernel has several s_endpgm: early returns)
    print(lines[i0].split(":")[0])
    meta = {}
    for ln in lines[i1:i1 + 80]:
        m = re.match(r"\s*[;.]\s*\.?(NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|vgpr_count|sgpr_count):?\s*(\d+)", ln)
        if m:
            meta[m.group(1)] = int(m.group(2))
    print("  ", meta)
    # basic blocks
    blocks, cur = [], {"label": "entry", "ins": [], "depth": 0, "header": False}
    for ln in lines[i0 + 1:i1 + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln)
        if m:
            blocks.append(cur)
            cmt = m.group(2) or ""
            d = re.search(r"Depth=(\d+)", cmt)
            cur = {"label": m.group(1), "ins": [], "depth": int(d.group(1)) if d else 0, "header": "Loop Header" in cmt, "inner": "Inner Loop" in cmt, "cmt": cmt}
            continue
        s = ln.strip()
        if s.startswith(";") and not cur["ins"] and "cmt" in cur:  # the label's comment goes on over several lines (Parent Loop ... / => This Inner Loop Header: Depth=n)
            cur["cmt"] += " " + s
            d = re.search(r"This (Inner )?Loop Header: Depth=(\d+)", s)
            if d:
                cur["header"], cur["depth"], cur["inner"] = True, int(d.group(2)), bool(d.group(1))
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        cur["ins"].append(s.split(";")[0].strip())
    blocks.append(cur)
    # loops: a header block and the blocks that follow it until (and including) the one that branches back to it
    idx = {b["label"]: k for k, b in enumerate(blocks)}
    loops = []
    for k, b in enumerate(blocks):
        if not b.get("header"):
            continue
        last = k
        for j in range(k, len(blocks)):
            if any(re.search(r"s_cbranch\w*\s+" + re.escape(b["label"]) + r"\b|s_branch\s+" + re.escape(b["label"]) + r"\b", ins) for ins in blocks[j]["ins"]):
                last = j
        loops.append((k, last))

    def summarise(bs):
        h = collections.Counter()
        ops = collections.Counter()
        for b in bs:
            for ins in b["ins"]:
                op = ins.split()[0]
                h[classify(op)] += 1
                ops[op] += 1
        valu = sum(v for c, v in h.items() if c in ("vop2_fast", "fma", "half", "vop3_int", "mix", "pk", "trans", "f64", "cmp", "dpp"))
        cyc_valu = sum(COST[c] * v for c, v in h.items() if c in ("vop2_fast", "fma", "half", "vop3_int", "mix", "pk", "trans", "f64", "cmp", "dpp"))
        cyc_all = sum(COST[c] * v for c, v in h.items())
        return h, ops, valu, cyc_valu, cyc_all

    for k, last in loops:
        bs = blocks[k:last + 1]
        h, ops, valu, cv, ca = summarise(bs)
        inner = [1 for (k2, l2) in loops if k2 > k and l2 <= last]
        print("loop %s (depth %d, %d blocks%s): VALU %d (%.1f cyc, mean %.2f), SALU %d, LDS %d, VMEM %d, SMEM %d, branches %d, waits %d; all issue %.1f cyc"
              % (blocks[k]["label"], blocks[k]["depth"], len(bs), ", contains %d inner loops" % len(inner) if inner else "", valu, cv, cv / max(valu, 1), h["salu"], h["lds"], h["vmem"], h["smem"],
                 h["branch"], h["wait"], ca))
        print("     classes:", dict(sorted(h.items())))
        if a.dump:
            for b in bs:
                print("   ", b["label"] + ":")
                for ins in b["ins"]:
                    print("        ", ins)
    if a.blocks:
        for b in blocks:
            h, ops, valu, cv, ca = summarise([b])
            print("block %-12s depth %d  n=%3d VALU %3d SALU %3d LDS %2d VMEM %2d" % (b["label"], b.get("depth", 0), len(b["ins"]), valu, h["salu"], h["lds"], h["vmem"]))
    h, ops, valu, cv, ca = summarise(blocks)
    print("whole kernel: %d instructions, VALU %d, SALU %d, LDS %d, VMEM %d" % (sum(h.values()), valu, h["salu"], h["lds"], h["vmem"]))


if __name__ == "__main__":
    main()
