#!/bin/bash
# One rocprofv3 counter pass over a bench workload; prints per-launch sums for the dominant kernel.
# usage: bash tools/pmc_pass.sh <workload> <kernel substring> COUNTER [COUNTER ...]   (at most 8 SQ counters per pass)
wl=$1; kern=$2; shift 2
out=gpurun_out/pmc_$wl
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out.log 2>&1
python3 - "$out" "$kern" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
for d, c in list(acc.items())[-1:]:
    for k in sorted(c):
        print(f"{k:28s} {c[k]:.4g}")
PY
