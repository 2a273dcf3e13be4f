"""All dispatches of one kernel in a rocprofv3 --kernel-trace CSV, in start order: start, duration, queue, how many
of them are in flight at its start, and the busy fraction (union of the intervals / span) over windows of 20.
usage: python tools/chain_timeline.py <kernel_trace.csv> [kernel name substring = k_compress<1]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else "k_compress<1"
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows if sub in r["Kernel_Name"]]
ks.sort()
t0 = ks[0][0]
for i, (s, e, q) in enumerate(ks):
    live = sum(1 for (s2, e2, _) in ks[:i] if e2 > s)
    nxt = (ks[i + 1][0] - s) / 1e3 if i + 1 < len(ks) else 0.0
    print("%10.1f us  dur %7.1f  q%-3s in flight at start %d   next start +%6.1f" % ((s - t0) / 1e3, (e - s) / 1e3, q, live, nxt))
