#!/bin/bash
# Everything kept under profiles/<round>/ for one part of the workloads: bench lines, rocprofv3 kernel stats and
# counter passes, kernel-bench texts.  The box's time limit per call is 20 minutes, so the round is taken in parts.
# usage (GPU box, repository root): bash tools/round_profiles.sh <round> <part>     part: a | b | c
rnd=$1; part=$2
mkdir -p gpurun_out/$rnd profiles/$rnd
case $part in
a)  bash tools/round_bench.sh $rnd c3 c2 c4 c5 || exit 1
    for wl in c3 c4 c5; do bash tools/collect_profiles.sh $wl profiles/$rnd || exit 1; done ;;
b)  bash tools/round_bench.sh $rnd f1 p1 || exit 1
    for wl in c2 f1 p1; do bash tools/collect_profiles.sh $wl profiles/$rnd || exit 1; done ;;
c)  for k in c5 pcar pcas husmall pca1m fps render png; do
        timeout -k 10 240 python3 tools/kbench.py $k > gpurun_out/$rnd/kbench_$k.txt 2>&1 || exit 1
        tail -3 gpurun_out/$rnd/kbench_$k.txt
    done
    timeout -k 10 400 python3 tools/ingest_bench.py > gpurun_out/$rnd/ingest_e2e.json 2> gpurun_out/$rnd/ingest_e2e.err || exit 1
    tail -1 gpurun_out/$rnd/ingest_e2e.json ;;
esac
