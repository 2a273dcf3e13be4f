#!/bin/bash
# Per-kernel average times of the large tier for one chunk length: FLEN (8192 ... 131072), NF frames (10 M samples by default).
# usage (GPU box): FLEN=8192 bash tools/large_len_kstats.sh
F=${FLEN:-8192}; NFR=${NF:-$((10485760 / F))}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
out=$R/gpurun_out/llk_$F; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
FLEN=$F NF=$NFR rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $R/tools/large_trace.py > $out/run.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" $F $NFR <<'PY'
import csv, sys
print("frames of %s samples x %s" % (sys.argv[2], sys.argv[3]))
for r in csv.DictReader(open(sys.argv[1])):
    print("  %-56s calls %3s  avg %8.1f us" % (r["Name"].replace("atsc::", "").split("(")[0][:56], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
