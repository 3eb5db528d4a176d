#!/bin/bash
# A/B of MLGGD_DP_STOPEV (events for the communication stream riding on the producing launch) in the 1-rank rehearsal, per exchange mode
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-44s dp_mode %-9s %8.2f us/step   windows %.3f .. %.3f ms' % (sys.argv[1], d['config']['dp_mode'], d['ms_per_step']*1e3, d['timing']['window_ms_min'], d['timing']['window_ms_max']))" "$1"; }
for rep in 1 2; do
for ml in 0 1; do
  for m in shard gather; do
    for f in 0 1; do
      [ $m = shard ] && [ $f = 1 ] && continue
      extra=""; [ $ml = 1 ] && extra="--loss ml"
      MLGGD_DP_STOPEV=$ml MLGGD_DP_FINE=$f python3 $R/bench.py --rehearse-dp --dp-mode $m --steps 20 --warmup 5 --no-ml --no-dp-arms --no-kernel-timing 2>/dev/null | line "STOPEV=$ml $m DP_FINE=$f"
    done
  done
done
done
