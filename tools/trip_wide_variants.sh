#!/bin/bash
# k_large_trip243's wide form at other widths (dev builds: ATSC_BUILD_VARIANT=w768 ATSC_BUILD_DEFS=-DATSC_TRIP_WIDE=768 python -m
# atsc_amd.build), forced on (ATSC_TRIP_WIDE_MAX=1000000): average kernel time, encoder and decoder, NF frames of 131072 samples.
# usage (GPU box): bash tools/trip_wide_variants.sh "32 80" "'' w768 w1024"
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for nf in ${1:-32 80}; do
  for v in ${2:-"" w768 w1024}; do
    out=$R/gpurun_out/twv_${nf}_$v; rm -rf $out
    NF=$nf KLASS=mix ATSC_LIB_VARIANT=$v ATSC_TRIP_WIDE_MAX=1000000 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $R/tools/large_decode_trace.py > /dev/null 2>&1
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    python3 - "$f" $nf "${v:-w512}" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "trip243" in r["Name"]:
        print("NF %4s  %-6s %-42s calls %3s avg %8.1f us" % (sys.argv[2], sys.argv[3], r["Name"].replace("atsc::", "").split("(")[0][5:47], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
