"""profiles/frame_timeline.py <dir with *_kernel_trace.csv> [frame index]: the launches of one frame, start and duration in us."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_shade_all" in r["Kernel_Name"] or "k_resolve" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 2
a, b = ends[k - 1] + 1, ends[k] + 1
t0 = int(rows[a]["Start_Timestamp"])
prev = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    print("%8.1f us  gap %5.1f  dur %7.1f us  grid %8s  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r["Grid_Size_X"], name[:50]))
    prev = e
print("frame: %.1f us" % ((prev - t0) / 1e3))
