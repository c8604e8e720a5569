#!/bin/bash
# Per-kernel times (rocprofv3 --kernel-trace --stats) of one tools/kbench.py case.
# usage (GPU box, repository root): bash tools/kprof.sh <case>
out=gpurun_out/kprof_$1
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/kbench.py $1 > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, sys, re
from pathlib import Path
for f in sorted(Path(sys.argv[1]).rglob("*kernel_stats.csv"))[:1]:
    for r in list(csv.DictReader(open(f)))[:18]:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:70]
        print(f"{name:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
