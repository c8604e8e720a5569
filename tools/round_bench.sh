#!/bin/bash
# One bench line per workload into gpurun_out/<round>/ (copied to profiles/<round>/ afterwards).
# usage (GPU box, repository root): bash tools/round_bench.sh <round> [workload ...]
set -e
rnd=$1; shift
wls=${@:-c3 c2 c4 c5}
mkdir -p gpurun_out/$rnd
for wl in $wls; do
    timeout -k 10 400 python3 bench.py --workload $wl > gpurun_out/$rnd/bench_$wl.json 2> gpurun_out/$rnd/bench_$wl.err
    echo "$wl: $(cut -c1-400 gpurun_out/$rnd/bench_$wl.json)"
done
