"""Timeline of the LAST batch in a rocprofv3 --kernel-trace CSV: start (relative), duration and the gap to the previous
kernel's end, per dispatch.  usage: python tools/trace_timeline.py <kernel_trace.csv> [first kernel name substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else None
start_idx = 0
if first:
    idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    if idx:
        start_idx = idx[-1]
rows = rows[start_idx:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("atsc::", "").split("(")[0][:44]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f us  +%7.1f us  gap %6.1f  q%-3s %s  grid %s wg %s vgpr %s sgpr %s lds %s scratch %s" % (
        (s - t0) / 1e3, (e - s) / 1e3, gap, r.get("Queue_Id", "?"), name, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"),
        r.get("VGPR_Count", "?"), r.get("SGPR_Count", "?"), r.get("LDS_Block_Size", "?"), r.get("Scratch_Size", r.get("Private_Segment_Size", "?"))))
    prev_end = e if prev_end is None else max(prev_end, e)
print("total %.1f us" % ((prev_end - t0) / 1e3))
