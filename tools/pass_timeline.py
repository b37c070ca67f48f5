"""Who is on the GPU when: from a rocprofv3 --kernel-trace CSV directory of a bench.py run, the share of the timed region's
wall time during which (a) an audio first pass, (b) a fused EEG kernel, (c) only small kernels, (d) nothing is running.
python tools/pass_timeline.py <dir> [last_ms=200]"""
import csv, glob, sys
d = sys.argv[1]
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "")
        kind = "audio" if n.startswith("rips_cloud_kernel<512, 1, unsigned int") else "eeg" if n.startswith("eeg_window_kernel<3, false, 1, false") else \
               "small" if n.startswith(("rips_", "eeg_", "wasserstein", "diagram", "tau_", "recording_rows", "retry_collect")) else "other"
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, n[:40]))
ours = [e for e in ev if e[2] != "other"]
t_end = max(e[1] for e in ours)
t0 = t_end - int(last_ms * 1e6)
pts = []
for s, e, k, _ in ours:
    if e <= t0:
        continue
    pts.append((max(s, t0), +1, k)); pts.append((e, -1, k))
pts.sort()
cnt = {"audio": 0, "eeg": 0, "small": 0}
acc = {"audio only": 0, "eeg only": 0, "audio+eeg": 0, "small only": 0, "idle": 0}
prev = t0
for t, dlt, k in pts:
    span = t - prev
    if span > 0:
        a, g, sm = cnt["audio"] > 0, cnt["eeg"] > 0, cnt["small"] > 0
        key = "audio+eeg" if a and g else "audio only" if a else "eeg only" if g else "small only" if sm else "idle"
        acc[key] += span
    cnt[k] += dlt
    prev = t
tot = sum(acc.values())
print(f"last {last_ms:.0f} ms of product kernels:", ", ".join(f"{k} {100.0 * v / tot:.1f} %" for k, v in acc.items()))
