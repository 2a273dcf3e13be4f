#!/bin/bash
# SQ counters per kernel of the large tier (tools/large_trace.py, NF frames): usage tools/pmc_large.sh <out dir> [NF]
out=${1:-gpurun_out/pmc_large}; export NF=${2:-80}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
i=0
for set in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pass$i -- python3 tools/large_trace.py > /dev/null 2>&1 || echo "pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/pass*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('atsc::', '')[:34]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    if not k.startswith('k_large') and not k.startswith('k_compress_large'):
        continue
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    wc = max(c.get('SQ_WAVE_CYCLES', 1), 1)
    print("%-34s waves %7.0f  valu/wave %6.0f salu/wave %6.0f lds/wave %5.0f | active %.2f wait_inst %.2f wait_any %.2f | lds conflict %.2f of lds active, lds active/wave_cycles %.2f" % (
        k, c.get('SQ_WAVES', 0), c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_WAVES', 1), 1), c.get('SQ_INSTS_SALU', 0) / max(c.get('SQ_WAVES', 1), 1),
        c.get('SQ_INSTS_LDS', 0) / max(c.get('SQ_WAVES', 1), 1), c.get('SQ_ACTIVE_INST_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc,
        c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1), c.get('SQ_LDS_IDX_ACTIVE', 0) / wc))
PY
