"""per-launch durations and gaps of the last scan of a rocprofv3 kernel trace (B1 run): python tools/diag/gn_trace.py <dir>"""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "?")), (re.search(r"(\w+_kernel)", r["Kernel_Name"]) or re.search(r"(\w+)", r["Kernel_Name"])).group(1)))
rows.sort()
# last step: from the last org_count
idx = [i for i, r in enumerate(rows) if "org_count" in r[3]]
lo = idx[-2] if len(idx) > 1 else 0
hi = idx[-1]
prev_end = None
tot = 0
for s, e, q, n in rows[lo:hi]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{n:32s} q {q} dur {(e - s) / 1e3:7.2f} us  gap {gap:6.2f} us")
    prev_end = e
print("step wall", (rows[hi][0] - rows[lo][0]) / 1e3, "us")
