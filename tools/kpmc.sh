#!/bin/bash
# One rocprofv3 counter pass over a tools/kbench.py case; prints the sums over the last launch of a kernel.
# usage: bash tools/kpmc.sh <case> <kernel substring> COUNTER [COUNTER ...]   (at most 8 SQ counters per pass)
kase=$1; kern=$2; shift 2
out=gpurun_out/kpmc_$kase
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 tools/kbench.py $kase > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
grep '^hu\|^pca' $out.log
python3 - "$out" "$kern" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
for d in sorted(acc)[-1:]:
    for k in sorted(acc[d]):
        print(f"{k:28s} {acc[d][k]:.4g}")
PY
