#!/bin/bash
# Per-kernel average times of the large tier (configs[3] shape) from a rocprofv3 kernel trace.
# usage (GPU box): bash tools/large_kstats.sh <outdir>
out=${1:-gpurun_out/lk}; mkdir -p $out
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o t -- python3 $R/tools/large_c3_trace.py > $R/$out/run.log 2>&1
cd $R
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    print("%-60s calls %4s  avg %9.1f us  total %9.1f us" % (r["Name"].replace("atsc::", "").split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
t=$(find $out -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_timeline.py $t | tail -14
