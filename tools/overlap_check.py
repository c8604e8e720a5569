"""Reads a rocprofv3 kernel trace (csv) and prints, for the last step of a c5 bench run, when each kernel ran
relative to the image kernel: do the cloud kernels of the second stream run beside it?
usage: python tools/overlap_check.py <dir with *_kernel_trace.csv>"""
import csv, sys, re
from pathlib import Path
rows = []
for f in Path(sys.argv[1]).rglob("*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hu = [r for r in rows if "hu_band" in r["Kernel_Name"]]
h = hu[-1]
h0, h1 = int(h["Start_Timestamp"]), int(h["End_Timestamp"])
print(f"image kernel: {(h1 - h0) / 1e3:.1f} us, queue {h.get('Queue_Id')}")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < h0 - 400_000 or s > h1 + 400_000:
        continue
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:44]
    print(f"{name:44s} q{r.get('Queue_Id')} start {(s - h0) / 1e3:8.1f} us  end {(e - h0) / 1e3:8.1f} us")
