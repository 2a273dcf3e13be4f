"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE counter_collection CSVs) next
to the kernel durations of a --kernel-trace --stats pass.  gfx950: FETCH_SIZE is doubled (128-B requests tallied at
64 B, MI355X_MICROARCH.md); counter unit KiB.  usage: pmc_traffic.py <stats dir> <fetch dir> <write dir> [launches to skip]"""
import collections, csv, glob, os, sys
def counters(d, name):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(d + '/*/*counter_collection.csv'), key=os.path.getmtime)[-1:]:  # the newest run only
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name:
                agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    return agg
def durations(d):
    out = {}
    for f in sorted(glob.glob(d + '/*/*kernel_stats.csv'), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            out[r['Name'].split('(')[0]] = (int(r['Calls']), float(r['AverageNs']) / 1e3)
    return out
st, fe, wr = sys.argv[1:4]
F = counters(fe, 'FETCH_SIZE'); W = counters(wr, 'WRITE_SIZE'); D = durations(st)
tot_f = tot_w = tot_us = 0.0
print("%-46s %6s %9s %10s %10s %8s" % ("kernel", "calls", "avg us", "read MB", "write MB", "GB/s"))
for k in sorted(D, key=lambda k: -D[k][1] * D[k][0]):
    calls, us = D[k]
    f = 2 * 1024 * sum(F.get(k, [0])) / max(len(F.get(k, [1])), 1) / 1e6
    w = 1024 * sum(W.get(k, [0])) / max(len(W.get(k, [1])), 1) / 1e6
    print("%-46s %6d %9.1f %10.1f %10.1f %8.0f" % (k.replace('void ', '').replace('atsc::', '')[:46], calls, us, f, w, (f + w) * 1e6 / (us * 1e-6) / 1e9 if us else 0))
