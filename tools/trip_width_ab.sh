#!/bin/bash
# A/B of k_large_trip243's two workgroup widths (192 / 512 threads) over launch sizes: average kernel time from a rocprofv3
# kernel trace of tools/large_trace.py (encoder) and tools/large_decode_trace.py (decoder), NF frames of 131072 samples.
# usage (GPU box): bash tools/trip_width_ab.sh "32 80 128 256"
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for nf in ${1:-32 80 128 256}; do
  for wm in 0 1000000; do
    for t in large_trace large_decode_trace; do
      out=$R/gpurun_out/tw_${nf}_${wm}_$t; rm -rf $out
      NF=$nf KLASS=mix ATSC_TRIP_WIDE_MAX=$wm rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $R/tools/$t.py > /dev/null 2>&1
      f=$(find $out -name "*kernel_stats.csv" | head -1)
      python3 - "$f" $nf $wm <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "trip243" in r["Name"]:
        print("NF %4s  %s  %-40s calls %3s avg %8.1f us" % (sys.argv[2], "wide  " if int(sys.argv[3]) else "narrow", r["Name"].replace("atsc::", "").split("(")[0][5:45], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    done
  done
done
