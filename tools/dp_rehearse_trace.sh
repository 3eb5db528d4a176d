#!/bin/bash
# kernel-trace timeline of a few steady-state steps of one exchange mode in the 1-rank rehearsal (see dp_rehearse_ab.sh).
# NOTE: under rocprofv3 every dispatch is serialised across queues, so the overlap between the communication stream's
# copies and the GEMM kernels is NOT what an un-profiled run does; use it for order and per-kernel durations only.
# usage: bash tools/dp_rehearse_trace.sh <mode> [extra env assignments...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; M=${1:-shard}; shift
O=$R/gpurun_out/r04_dp_trace_$M; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/bench.py --rehearse-dp --dp-mode $M --steps 20 --warmup 5 --windows 2 --no-ml --no-dp-arms --no-kernel-timing > $O/bench.json 2> $O/bench.err
cd $R
python3 tools/dp_trace_timeline.py $O/trace > $O/timeline.txt 2>&1
rm -rf $O/trace
tail -80 $O/timeline.txt
