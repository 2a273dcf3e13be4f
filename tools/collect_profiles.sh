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
for c in 1 4; do echo "== bench, $c chain(s)"; python3 bench.py --chains $c --no-cpu-baseline --no-end-to-end > $G/${tag}_bench_chains$c.json 2>/dev/null; done
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
echo "== chain stamps"; for c in 1 2; do CHAINS=$c ATSC_LIB_VARIANT=stamps python3 tools/chain_stamp_probe.py > $G/${tag}_chain_stamps_$c.txt 2>&1; done
echo "== resident experiment"; for r in 0 1; do for c in 1 2; do if [ $r = 1 ]; then export ATSC_RESIDENT=1; else unset ATSC_RESIDENT; fi; echo "resident $r"; ATSC_LIB_VARIANT= BRIEF=1 STEPS=200 CHAINS=$c python3 tools/chain_stamp_probe.py 2>&1 | grep chains; done; done > $G/${tag}_resident_ab.txt 2>&1; unset ATSC_RESIDENT
echo "== dispatch probe"; ./tools/dispatch_probe > $G/${tag}_dispatch_probe.txt 2>&1
echo "== gap probe"; ./tools/gap_probe > $G/${tag}_gap_probe.txt 2>&1
echo "== host path"; python3 tools/host_path_probe.py > $G/${tag}_host_path.txt 2>&1
echo "== large decode"; python3 tools/large_decode_probe.py > $G/${tag}_large_decode_probe.txt 2>&1
NF=80 KLASS=mix bash tools/large_dkstats.sh $G/${tag}_ldk > $G/${tag}_large80_decode_kernels.txt 2>&1
echo "== large ties"; python3 tools/large_tie_fraction.py > $G/${tag}_large_ties.txt 2>&1
echo "== queue probe"; python3 tools/queue_pipe_probe.py > $G/${tag}_queue_probe.txt 2>&1
echo "== config3 full"; python3 bench.py --workload config3 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > $G/${tag}_config3_full_1gpu.txt 2>&1
echo "== share2"; ATSC_BENCH_SHARE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --series 128 --steps 5 --warmup 2 > $G/${tag}_bench_share2.json 2>/dev/null
echo done
