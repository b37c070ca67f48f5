#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/collect_profiles.sh:
   <out>/<tag>_kernel_stats.csv   per-kernel calls / total / average / percentage (from --stats; five passes in flight)
   <out>/<tag>_kernel_stats_lanes1.csv   the same with one pass in flight, eager launches (per-launch times fit the step)
   <out>/<tag>_pmc_summary.json   FETCH_SIZE / WRITE_SIZE per launch, averaged per kernel name
   <out>/traffic.json             HBM bytes per launch of the first-pass kernel of each bench stage
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(out, d, "**", pat), recursive=True))
    return hits[0] if hits else None


def short(name):
    name = re.sub(r"\(.*$", "", name).strip()
    return name.replace("void ", "")


for d, suffix in (("prof_stats", "kernel_stats"), ("prof_stats1", "kernel_stats_lanes1")):
    stats = find(d, "*kernel_stats.csv")
    if not stats:
        continue
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(out, f"{tag}_{suffix}.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent", "min_ns", "max_ns"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r.get("MinNs", ""), r.get("MaxNs", "")])
    print(suffix, ":", len(rows), "kernels")

pmc = {}
for d, ctr in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
    path = find(d, "*counter_collection.csv")
    if not path:
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != ctr:
            continue
        k = short(r["Kernel_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    for k, (s, n) in acc.items():
        e = pmc.setdefault(k, {})
        e[f"{ctr}_KB_avg_per_launch"] = round(s / n, 2)
        e["launches_seen"] = n
json.dump(pmc, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1)

STAGE_KERNEL = {
    "eeg_window": "eeg_window_kernel<3, false, 1, false", "corr_dist": "corr_dist_kernel",
    "rips_eeg": "rips_dm_kernel<256, 1, 1, unsigned long long, false", "rips_audio": "rips_cloud_kernel<512, 1, unsigned int, false",
    "wasserstein_h0": "wasserstein_kernel<2", "wasserstein_h1": "wasserstein_kernel<4", "tau": "tau_kernel",
    "finish": "diagram_finish_kernel",
}
# gfx950: FETCH_SIZE reports half of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md).  The window
# fetch of corr_dist / the fused EEG kernel reads 512 contiguous bytes per wave instruction and shows exactly that;
# the other kernels gather 8 B per lane from short rows and are left as counted.
FETCH_X2 = {"corr_dist", "eeg_window"}
traffic = {}
launch_windows = None
for stage, pat in STAGE_KERNEL.items():
    for k, e in pmc.items():
        if k.startswith(pat) or pat in k:
            fetch = e.get("FETCH_SIZE_KB_avg_per_launch", 0.0) * (2.0 if stage in FETCH_X2 else 1.0)
            traffic[stage] = int(round((fetch + e.get("WRITE_SIZE_KB_avg_per_launch", 0.0)) * 1024))
            break
traffic["_windows_per_launch"] = 3540
traffic["_note"] = ("HBM bytes per launch of the first-pass kernel of each stage, at 3,540 windows per launch (the reduced corpus "
                    "of tools/collect_profiles.sh): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB -> "
                    "bytes).  gfx950 correction: FETCH_SIZE doubled for the window fetch of corr_dist / eeg_window (wide "
                    "coalesced streaming reads are tallied at half their size); the other kernels gather 8 B per lane and "
                    "are left as counted.  bench.py scales the figure to its own windows per launch.")
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
