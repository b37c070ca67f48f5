"""Print the kernel / copy timeline of a rocprofv3 --kernel-trace --memory-copy-trace run (CSV output directory).
python tools/timeline.py <dir> [first_fraction] [n_rows]"""
import csv, glob, sys
d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 80
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), r["Kernel_Name"].replace("void ", "")[:44]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "?")[-16:]))
ev.sort()
t0 = ev[0][0]
for s, e, q, n in ev[int(len(ev) * frac):][:nrows]:
    print(f"{(s - t0) / 1e6:10.3f} +{(e - s) / 1e6:8.3f} ms  {q:>5}  {n}")
