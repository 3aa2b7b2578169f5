"""HBM traffic per launch of the headline kernel, per launch shape, from the PMC passes of tools/prof.sh:
tools/pmc_traffic.py <gpurun_out/prof_TAG> <config>  ->  the <config> entry of profiles/<round>_pmc_traffic.json on stdout.

FETCH_SIZE and WRITE_SIZE are collected in separate rocprofv3 runs (they do not fit one pass: MI355X_MICROARCH.md, rocprofv3
PMC slots) and are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled
(same guide, HBM section); WRITE_SIZE is taken as it is.  Only the production (non-counting) instance of the march kernel is
read, and only launches of the shape the run's JSON line reports (the warm-up's padded launch has the same shape)."""
import csv
import glob
import json
import os
import sys

root, config = sys.argv[1], sys.argv[2]
KERNEL = "raymarch_naive_kernel" if config == "c2" else "raymarch_staged_kernel"


def counter_mean(d, name):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if KERNEL in k and r["Counter_Name"] == name and "true>(vk::LaunchDesc" not in k.replace(" ", ""):
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


out = {"per_frames_per_launch": {}}
for shape in sorted(os.listdir(root)):
    d = os.path.join(root, shape)
    line = os.path.join(d, "bench_line_under_rocprof.json")
    if not os.path.isdir(d) or not os.path.exists(line) or shape not in ("default", "driver"):  # (the headline's two launch shapes; other cases have their own summaries)
        continue
    try:
        j = json.loads(open(line).read())
    except Exception:
        continue
    fpl = j["frames_per_launch"]
    fetch, nf = counter_mean(os.path.join(d, "fetch"), "FETCH_SIZE")
    write, nw = counter_mean(os.path.join(d, "write"), "WRITE_SIZE")
    if fetch is None or write is None:
        continue
    stats = {}
    ks = os.path.join(d, "kernel_stats.csv")
    if os.path.exists(ks):
        for r in csv.DictReader(open(ks)):
            if KERNEL in r["Name"] and "true>(vk::LaunchDesc" not in r["Name"].replace(" ", ""):
                stats = {"kernel_avg_ns_rocprofv3": float(r["AverageNs"]), "kernel_min_ns_rocprofv3": float(r["MinNs"]), "kernel_calls_rocprofv3": int(r["Calls"])}
    out["per_frames_per_launch"][str(fpl)] = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/prof.sh, shape '%s': `python3 bench.py --no-extras --no-cpu-baseline --headline-only%s`), "
                  "means over the run's %d / %d launches of the production %s instance" % (shape, "" if shape == "default" else " --steps 20 --warmup 5", nf, nw, KERNEL),
        "frames_per_launch": fpl, "cameras": j["config"].get("cameras", ""),
        "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
        "correction": "gfx950: FETCH_SIZE x 2 (wide coalesced reads are reported at half their bytes), WRITE_SIZE as reported; KiB -> bytes x 1024",
        "hbm_bytes_per_launch": int(2 * fetch * 1024 + write * 1024), "hbm_bytes_per_frame": int((2 * fetch * 1024 + write * 1024) / fpl),
        "launch_ms_by_hip_events_same_run": j.get("roofline", {}).get("launch_ms"), **stats}
print(json.dumps(out, indent=1))
