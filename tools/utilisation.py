#!/usr/bin/env python3
"""Issue utilisation of the march kernels from the PMC passes of tools/prof.sh: for each profiled case

    cycles per SIMD      = GRBM_GUI_ACTIVE / 8                      (the counter sums the 8 XCDs; MI355X_MICROARCH.md, DVFS)
    cycles per VALU inst = cycles per SIMD / (SQ_INSTS_VALU / 1024 SIMDs)
    VALU pipe occupancy  = SQ_INSTS_VALU / 1024 x (mean issue cost of the kernel's hot loop) / cycles per SIMD
    issue-slot occupancy = (VALU + SALU + LDS + VMEM + SMEM instructions) / 1024 x 2.4 / cycles per SIMD

with the mean issue cost from the hot loop's static instruction mix (tools/isa_hist.py) priced by the measured classes of
profiles/r03_ubench_valu_issue_rate.txt (2 / 4 / 8 cycles), and 2.4 cycles per instruction of any kind as the issue-slot price
(v_fma + s_add pairs: 4.8).  The two occupancies bound the kernel from two sides (pipe cycles, issue slots); a kernel is at its
issue limit when either is near 1.

  python tools/utilisation.py gpurun_out/prof_r04 --json=profiles/r04_utilisation.json > profiles/r04_utilisation.txt
"""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# case -> (directory under the given roots, kernel-name fragment in the counter csv, mangled-name regex for isa_hist, VALU count of the hot loop to price)
# (kernel template arguments: raymarch_naive_kernel<VOL, SKIP, SAFE, WALK, AHEAD, OUT, COUNT>; raymarch_compute_records_kernel<OUT, COUNT, SKIP, RING, REV>)
CASES = [
    ("C2 headline: skip march (bit-exact walk), 128 orbit frames per launch", "default", "raymarch_naive_kernel<3, true, false, 0, false, 1, false>", "raymarch_naive_kernelILi3ELb1ELb0ELi0ELb0ELi1ELb0E", None, None),
    ("C2 in tolerance mode (VK_RENDER_FAST_WALK), 128 orbit frames per launch", "fastwalk", "raymarch_naive_kernel<3, true, false, 2, false, 1, false>", "raymarch_naive_kernelILi3ELb1ELb0ELi2ELb0ELi1ELb0E", None, None),
    ("C2 fog, dense march, 8 frames per launch", "fogbatch", "raymarch_naive_kernel<3, false, false, 0, false, 1, false>", "raymarch_naive_kernelILi3ELb0ELb0ELi0ELb0ELi1ELb0E", 88, None),
    ("C5: staged u8, 3840x2160, single frame (one window per 256-thread group)", "c5", "raymarch_staged_group_kernel<10, 1, false>", "raymarch_staged_group_kernelILi10ELi1ELb0E", 50, 2048 ** 3),
    ("C4: staged f16, 1920x1080, single frame", "c4", "raymarch_staged_kernel<11, 1, false>", "raymarch_staged_kernelILi11ELi1ELb0E", 50, 2 * 1024 ** 3),
    ("compute twin (16-byte records, exact skipping, 6-buffer request ring), xor 1280x720, single frame", "xor", "raymarch_compute_records_kernel<1, false, true, 6, 1>", "raymarch_compute_records_kernelILi1ELb0ELb1ELi6ELi1E", None, None),
    ("C3: procedural, 1920x1080, single frame", "c3", "raymarch_procedural_kernel<1, false, false>", "raymarch_procedural_kernelILi1ELb0ELb0E", None, None),
]


def counters(d, frag):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if frag in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = sum(v) / len(v)
    return out


def kernel_ns(d, frag):
    f = os.path.join(d, "kernel_stats.csv")
    if os.path.exists(f):
        rows = list(csv.DictReader(open(f)))
        for r in rows:  # the exact instance first (a counting or probe-ahead instance of the same kernel may have run more often)
            if frag in r["Name"]:
                return float(r["AverageNs"]), int(r["Calls"])
        for r in rows:
            if frag.split("<")[0] in r["Name"] and frag.split("<")[1][:8] in r["Name"]:
                return float(r["AverageNs"]), int(r["Calls"])
    return None, 0


def loop_mix(regex, want_valu):
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hist.py"), regex], capture_output=True, text=True).stdout
    best = None
    for m in re.finditer(r"loop (\S+) \(depth (\d+), (\d+) blocks[^)]*\): VALU (\d+) \(([\d.]+) cyc, mean ([\d.]+)\), SALU (\d+), LDS (\d+), VMEM (\d+)", txt):
        valu, mean = int(m.group(4)), float(m.group(6))
        if want_valu is not None and valu == want_valu:
            return valu, mean, m.group(1)
        if want_valu is None and (best is None or valu > best[0]):
            best = (valu, mean, m.group(1))
    return best if best else (0, 3.0, "?")


TAG = "r04"


def main():
    global TAG
    TAG = next((a[6:] for a in sys.argv[1:] if a.startswith("--tag=")), TAG)
    roots = [a for a in sys.argv[1:] if not a.startswith("--")]
    jpath = next((a[7:] for a in sys.argv[1:] if a.startswith("--json=")), None)
    jout = {}
    print(__doc__.split("  python tools")[0].strip())
    print()
    for title, sub, frag, regex, want, vol_bytes in CASES:
        d = next((os.path.join(r, sub) for r in roots if os.path.isdir(os.path.join(r, sub))), None)
        if d is None:
            continue
        c = counters(d, frag)
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        ns, calls = kernel_ns(d, frag)
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        valu = c["SQ_INSTS_VALU"] / 1024.0
        others = sum(c.get(k, 0.0) for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")) / 1024.0
        lv, mean, label = loop_mix(regex, want)
        jout[sub] = {"case": title, "kernel": frag, "kernel_ms_rocprofv3": (ns or 0) / 1e6, "cycles_per_valu_instruction": cyc / valu, "hot_loop_mean_issue_cycles": mean,
                     "valu_pipe_occupancy": valu * mean / cyc, "issue_slot_occupancy": (valu + others) * 2.4 / cyc,
                     "source": "profiles/%s_utilisation.txt (tools/utilisation.py over the PMC passes of tools/prof.sh; issue classes: profiles/r03_ubench_valu_issue_rate.txt)" % TAG}
        # physical HBM traffic of the launch: FETCH_SIZE (KiB, x 2 on gfx950: wide coalesced reads are reported at half their bytes) + WRITE_SIZE (KiB), separate passes
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c and ns:
            hbm = 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0
            jout[sub].update({"hbm_bytes_per_launch": hbm, "hbm_fetch_bytes_per_launch": 2.0 * c["FETCH_SIZE"] * 1024.0, "hbm_frac_of_peak": hbm / (ns * 1e-9) / 8.0e12})
            if vol_bytes:
                jout[sub]["volume_bytes"] = vol_bytes
                jout[sub]["refetch_factor"] = 2.0 * c["FETCH_SIZE"] * 1024.0 / vol_bytes
        print("== %s" % title)
        print("   kernel %s: %.4f ms mean over %d launches (rocprofv3 --kernel-trace), %.2f GHz by GRBM_GUI_ACTIVE / 8 / time" % (frag, (ns or 0) / 1e6, calls, cyc / (ns or 1)))
        print("   cycles per SIMD %.3e | VALU instructions per SIMD %.3e -> %.2f cycles per VALU instruction" % (cyc, valu, cyc / valu))
        print("   hot loop %s: %d VALU, mean issue cost %.2f cycles by the measured classes -> VALU pipe occupancy %.2f" % (label, lv, mean, valu * mean / cyc))
        print("   SALU %.3e, LDS %.3e, VMEM %.3e, SMEM %.3e per SIMD -> issue-slot occupancy (all instructions x 2.4) %.2f"
              % (c.get("SQ_INSTS_SALU", 0) / 1024, c.get("SQ_INSTS_LDS", 0) / 1024, (c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)) / 1024, c.get("SQ_INSTS_SMEM", 0) / 1024,
                 (valu + others) * 2.4 / cyc))
        if "SQ_WAVE_CYCLES" in c:
            print("   waves: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES %.2f (waiting for an issue slot), SQ_WAIT_ANY / SQ_WAVE_CYCLES %.2f (waiting on a counter), mean waves per SIMD %.1f"
                  % (c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAVE_CYCLES"] * 4.0 / 1024.0 / cyc))
        if "hbm_bytes_per_launch" in jout[sub]:
            j = jout[sub]
            print("   HBM (PMC, FETCH_SIZE x 2 + WRITE_SIZE): %.3f GB per launch = %.2f of 8 TB/s physical%s"
                  % (j["hbm_bytes_per_launch"] / 1e9, j["hbm_frac_of_peak"], (", the volume's %.2f GB fetched %.2f times" % (vol_bytes / 1e9, j["refetch_factor"])) if vol_bytes else ""))
        if c.get("SQ_LDS_IDX_ACTIVE"):
            print("   LDS: bank-conflict cycles / active cycles %.2f; LDS instructions per CU-cycle %.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], c.get("SQ_INSTS_LDS", 0) / 256.0 / cyc))
        print()
    if jpath:
        import json
        json.dump(jout, open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
