#!/bin/bash
# One profiling call for a round's committed summaries (GPU box): writes gpurun_out/<tag>_*; afterwards, here,
# `python tools/make_profiles.py <tag>` copies the summaries into profiles/.   usage: tools/collect_profiles.sh r02
set -u
tag=${1:-r02}
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp && cd "$root"
G=gpurun_out
mkdir -p $G
echo "== bench"; python3 bench.py > $G/${tag}_bench.json 2> $G/${tag}_bench.err
echo "== bench, four chains"; python3 bench.py --chains --no-cpu-baseline > $G/${tag}_bench_chains.json 2>/dev/null
echo "== kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $G/${tag}_stats -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end > $G/${tag}_stats_bench.json 2>/dev/null
echo "== pmc"; tools/pmc_collect.sh $G/${tag}_pmc --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $G/${tag}_pmc_summary.txt 2>&1
for nf in 80 256; do
  echo "== large $nf"
  NF=$nf rocprofv3 --kernel-trace --stats --output-format csv -d $G/${tag}_large$nf -- python3 tools/large_trace.py > /dev/null 2>&1
  NF=$nf rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $G/${tag}_large${nf}_fetch -- python3 tools/large_trace.py > /dev/null 2>&1
  NF=$nf rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $G/${tag}_large${nf}_write -- python3 tools/large_trace.py > /dev/null 2>&1
done
echo "== other configs"; python3 tools/bench_configs.py quick > $G/${tag}_other_configs.txt 2>&1
echo "== length probe"; python3 tools/length_probe.py > $G/${tag}_length_probe.txt 2>&1
echo "== stamps"; ATSC_LIB_VARIANT=stamps python3 tools/stamp_probe.py > $G/${tag}_stamps_256.txt 2>&1
for f in 2048 4096; do FLEN=$f ATSC_LIB_VARIANT=stamps python3 tools/stamp_probe.py > $G/${tag}_stamps_$f.txt 2>&1; done
echo "== queue probe"; python3 tools/queue_pipe_probe.py > $G/${tag}_queue_probe.txt 2>&1
echo "== config3 full"; python3 bench.py --workload config3 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > $G/${tag}_config3_full_1gpu.txt 2>&1
echo "== share2"; ATSC_BENCH_SHARE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --series 128 --steps 5 --warmup 2 > $G/${tag}_bench_share2.json 2>/dev/null
echo done
