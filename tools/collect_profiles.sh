#!/bin/bash
# rocprofv3 passes behind profiles/<round>/: kernel trace + stats, then one counter per run.
# usage (on the GPU box, from the repository root): bash tools/collect_profiles.sh <workload> <round_dir>
set -e
wl=$1; rdir=$2
out=gpurun_out/prof_$wl
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $out/stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
python3 tools/collect_profiles.py $wl $out $rdir
mkdir -p gpurun_out/$rdir && cp $rdir/${wl}_* gpurun_out/$rdir/ && cp profiles/traffic.json gpurun_out/traffic.json
