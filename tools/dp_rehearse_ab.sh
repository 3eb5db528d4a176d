#!/bin/bash
# The fixed cost of the data-parallel exchange path WITHOUT links: bench.py --rehearse-dp (a 1-rank RCCL communicator on
# one GPU) per exchange mode and for both factor-exchange granularities, plus RCCL's own kernels in a kernel trace.
# Output: gpurun_out/r03_dp_rehearse/ (copy the summary into profiles/).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r03_dp_rehearse; mkdir -p $O
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-34s dp_mode %-9s %8.2f us/step   windows %.3f .. %.3f ms' % (sys.argv[1], d['config']['dp_mode'], d['ms_per_step']*1e3, d['timing']['window_ms_min'], d['timing']['window_ms_max']))" "$1"; }
{
python3 $R/bench.py --steps 20 --warmup 5 --no-ml --no-kernel-timing --no-cpu-baseline 2>/dev/null | line "single GPU, no communicator"
for f in 1 0; do MLGGD_DP_FINE=$f python3 $R/bench.py --rehearse-dp --steps 20 --warmup 5 --no-ml --no-dp-arms --no-kernel-timing 2>/dev/null | line "gather, MLGGD_DP_FINE=$f"; done
for m in shard allreduce; do python3 $R/bench.py --rehearse-dp --dp-mode $m --steps 20 --warmup 5 --no-ml --no-dp-arms --no-kernel-timing 2>/dev/null | line "$m"; done
python3 $R/bench.py --rehearse-dp --loss ml --steps 20 --warmup 5 --no-dp-arms --no-kernel-timing 2>/dev/null | line "gather, ML-GGD loss (+ 257-float all-reduce)"
} > $O/modes.txt
cat $O/modes.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --rehearse-dp --steps 20 --warmup 5 --windows 3 --no-ml --no-kernel-timing > $O/profiled.json 2> $O/profiled.err
cd $R
cat $O/trace/*/*kernel_stats.csv | cut -c1-200 | head -24 > $O/kernel_stats_head.csv
cat $O/kernel_stats_head.csv
rm -rf $O/trace
