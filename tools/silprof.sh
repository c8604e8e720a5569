#!/bin/bash
# Per-kernel times of the silhouette stage (OpenCV branch) on 20 000 synthetic 224x224 views.
# usage (GPU box, repository root): bash tools/silprof.sh
out=gpurun_out/silprof
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/siltime.py ${1:-20000} > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, sys
from pathlib import Path
for f in sorted(Path(sys.argv[1]).rglob("*kernel_stats.csv"))[:1]:
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
