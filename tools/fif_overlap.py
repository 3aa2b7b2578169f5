"""Do single-frame launches on different streams overlap?  Reads a rocprofv3 --kernel-trace csv (kernel_trace.csv) and reports, for the
march kernel's last N launches: mean duration, mean start-to-start interval, and the share of the span during which 1 / 2 / 3+ launches
were running at once.  usage: tools/fif_overlap.py <kernel_trace.csv> [kernel-name-substring] [last N]"""
import csv, sys

path, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "raymarch")
last = int(sys.argv[3]) if len(sys.argv) > 3 else 64
rows = [r for r in csv.DictReader(open(path)) if sub in r["Kernel_Name"]]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows)[-last:]
dur = [b - a for a, b, _ in iv]
gaps = [iv[i + 1][0] - iv[i][0] for i in range(len(iv) - 1)]
ev = sorted([(a, 1) for a, _, _ in iv] + [(b, -1) for _, b, _ in iv])
depth, prev, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - prev)
    prev, depth = t, depth + d
span = iv[-1][1] - iv[0][0]
print("launches %d  queues %s" % (len(iv), sorted({q for _, _, q in iv})))
print("mean kernel duration  %.1f us" % (sum(dur) / len(dur) / 1e3))
print("mean start-to-start   %.1f us   (span / launches = %.1f us per frame)" % (sum(gaps) / len(gaps) / 1e3, span / len(iv) / 1e3))
for k in sorted(hist):
    if hist[k]:
        print("  %d launch(es) running: %5.1f %% of the span" % (k, 100.0 * hist[k] / span))
