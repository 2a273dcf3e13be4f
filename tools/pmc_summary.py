"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel (avg per launch)."""
import csv, glob, collections, sys
pat = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc*'
kern = sys.argv[2] if len(sys.argv) > 2 else 'k_compress'
for d in sorted(glob.glob(pat + '/')):
    for f in glob.glob(d + '*/*counter_collection.csv'):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in sorted(agg.items()):
            print("%-28s launches %d avg/launch %.4g" % (k, len(v), sum(v) / len(v)))
