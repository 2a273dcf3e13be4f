#!/bin/bash
# Round 4's profiling call (GPU box): writes gpurun_out/r04_*; afterwards, here, `python tools/make_profiles.py r04`.
#   usage: tools/collect_profiles_r04.sh [part]     part: all | bench | pmc | large | probes
set -u
tag=r04
part=${1:-all}
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp && cd "$root"
G=gpurun_out
mkdir -p $G
if [ $part = all ] || [ $part = bench ]; then
echo "== bench (default: 200 steps)"; python3 bench.py > $G/${tag}_bench.json 2> $G/${tag}_bench.err
echo "== bench (the driver's flags)"; python3 bench.py --steps 20 --warmup 5 > $G/${tag}_bench_driver.json 2> $G/${tag}_bench_driver.err
for c in 1 4; do echo "== bench, $c chain(s)"; python3 bench.py --chains $c --no-cpu-baseline --no-end-to-end > $G/${tag}_bench_chains$c.json 2>/dev/null; done
# kernel stats of a run whose headline launches do NOT overlap: one chain, plain single-stream calls
echo "== kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $G/${tag}_stats -- python3 bench.py --chains 1 --no-pipeline --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $G/${tag}_stats_bench.json 2>/dev/null
fi
if [ $part = all ] || [ $part = pmc ]; then
echo "== pmc"; tools/pmc_collect.sh $G/${tag}_pmc --chains 1 --no-pipeline --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $G/${tag}_pmc_summary.txt 2>&1
fi
if [ $part = all ] || [ $part = large ]; then
for nf in 80 256; do
  echo "== large $nf"
  NF=$nf rocprofv3 --kernel-trace --stats --output-format csv -d $G/${tag}_large$nf -- python3 tools/large_trace.py > /dev/null 2>&1
  NF=$nf rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $G/${tag}_large${nf}_fetch -- python3 tools/large_trace.py > /dev/null 2>&1
  NF=$nf rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $G/${tag}_large${nf}_write -- python3 tools/large_trace.py > /dev/null 2>&1
done
echo "== large decode 80"
NF=80 KLASS=mix rocprofv3 --kernel-trace --stats --output-format csv -d $G/${tag}_ldec80 -- python3 tools/large_decode_trace.py > /dev/null 2>&1
NF=80 KLASS=mix rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $G/${tag}_ldec80_fetch -- python3 tools/large_decode_trace.py > /dev/null 2>&1
NF=80 KLASS=mix rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $G/${tag}_ldec80_write -- python3 tools/large_decode_trace.py > /dev/null 2>&1
echo "== f256 decode"
rocprofv3 --kernel-trace --stats --output-format csv -d $G/${tag}_dec256 -- python3 tools/decode_trace.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $G/${tag}_dec256_fetch -- python3 tools/decode_trace.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $G/${tag}_dec256_write -- python3 tools/decode_trace.py > /dev/null 2>&1
for ns in 16 256; do NS=$ns bash tools/large_c3_kstats.sh $G/${tag}_c3k$ns > $G/${tag}_c3_kstats$ns.txt 2>&1; done
fi
if [ $part = all ] || [ $part = probes ]; then
echo "== other configs"; PAPER=1 python3 tools/bench_configs.py quick > $G/${tag}_other_configs.txt 2>&1
python3 tools/bench_configs.py > $G/${tag}_other_configs_256.txt 2>&1
echo "== length probe"; python3 tools/length_probe.py > $G/${tag}_length_probe.txt 2>&1
echo "== fixed cost"; python3 tools/fixed_cost_probe.py > $G/${tag}_fixed_cost.txt 2>&1
echo "== fast left"
{ echo "# 80 frames, e = 5 %"; NF=80 python3 tools/fast_left_probe.py; echo "# 256 frames, e = 5 %"; NF=256 python3 tools/fast_left_probe.py;
  echo "# 32 frames, e = 1 %"; ERR=1 NF=32 python3 tools/fast_left_probe.py; } > $G/${tag}_fast_left.txt 2>&1
echo "== stamps"; ATSC_LIB_VARIANT=stamps python3 tools/stamp_probe.py > $G/${tag}_stamps_256.txt 2>&1
echo "== large decode probe"; python3 tools/large_decode_probe.py > $G/${tag}_large_decode_probe.txt 2>&1
echo "== config3 full"; python3 bench.py --workload config3 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > $G/${tag}_config3_full_1gpu.txt 2>&1
echo "== share2"; ATSC_BENCH_SHARE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --series 128 --steps 5 --warmup 2 > $G/${tag}_bench_share2.json 2>/dev/null
echo "== fuzz soak"
{ python3 tools/fuzz_soak.py 500 560; FUZZ_LARGE=2 python3 tools/fuzz_soak.py 600 630; FUZZ_LARGE=1 python3 tools/fuzz_soak.py 700 715; python3 tools/fuzz_decode.py; } > $G/${tag}_fuzz_soak.txt 2>&1
fi
echo done
