"""Timeline of one replayed graph from a rocprofv3 kernel trace: python tools/trace_overlap.py <trace dir>
Prints, for the last complete run of kernels between two idle gaps, every kernel's start / end relative to the first start (us)."""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
# take a window of 24 kernels near the end of the trace
tailrows = rows[-60:-20]
t0 = tailrows[0][0]
for s, e, n in tailrows:
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:6.1f} us)  {short(n)[:60]}")
