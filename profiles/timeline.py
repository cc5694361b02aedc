"""Print the kernel timeline of the last full frame from a rocprofv3 --kernel-trace csv directory:
start offset, duration and gap to the previous kernel's end (all in microseconds).  An optional second argument picks an earlier frame (1 = the one before).  A frame starts at a
camera-ray launch (k_trace<0, true, ...> or k_bounce<..., true, ...>)."""
import csv, glob, os, sys

def main(d, back=0):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    starts = [i for i, n in enumerate(names) if "k_trace<0, true" in n or ("k_bounce<" in n and ", true, " in n.split("k_bounce<")[1][:12])]
    if len(starts) < 2 + back:
        print("no complete frame found"); return
    i0, i1 = starts[-2 - back], starts[-1 - back]
    t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
    for r in rows[i0:i1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print("%9.1f  dur %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:50]))
        prev_end = max(prev_end, e)
    print("frame span %.1f us" % ((prev_end - t0) / 1e3))

if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
