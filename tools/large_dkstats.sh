#!/bin/bash
# Per-kernel average times of the large tier's DECODER (80 frames of 131072 samples; NF=...) from a rocprofv3 kernel trace.
out=${1:-gpurun_out/ldk}; mkdir -p $out
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o t -- python3 $R/tools/large_decode_trace.py > $R/$out/run.log 2>&1
cd $R
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    print("%-60s calls %4s  avg %9.1f us  total %9.1f us" % (r["Name"].replace("atsc::", "").split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
