"""Turns rocprofv3 output directories into the summaries kept under profiles/.

Run on the GPU box after the rocprofv3 passes of tools/collect_profiles.sh:
    python3 tools/collect_profiles.py <workload> <out_dir> <round_dir>
reads  <out_dir>/stats/**/*kernel_stats.csv, <out_dir>/fetch/**/*counter_collection.csv and
<out_dir>/write/**/*counter_collection.csv, writes <round_dir>/<workload>_kernel_stats.csv and
<round_dir>/<workload>_pmc_hbm.json and updates profiles/traffic.json.
"""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
DOMINANT = {"c3": "hu_band_kernel", "c2": "hu_band_kernel", "c4": "pca_sets_full_kernel", "c5": "hu_band_kernel",
            "f1": "sil_mask_kernel", "p1": "render_views_kernel"}
UNITS = {"c3": (125000, 262144), "c2": (100000, 65536), "c4": (1000000, 12288), "c5": (64, 67108864), "f1": (10000, 50176),
         "p1": (2000, 12288)}
# FETCH_SIZE on gfx950 tallies the 128-byte requests of a 16 B/lane coalesced stream at 64 bytes, i.e. half the
# bytes; other request mixes are uncalibrated.  The factor is applied only to kernels whose loads are all
# 16 B/lane (hu_band_kernel: one global_load_dwordx4 per chunk; pca_sets_full_kernel: three
# float4 per lane), each of which reads its input exactly once, so the corrected figure can be checked against
# the algorithmic byte count (ratio recorded below).  Kernels with another request mix get no traffic figure.
FETCH_FACTOR = {"hu_band_kernel": 2.0, "pca_sets_full_kernel": 2.0}


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name


def counter_means(folder):
    """kernel -> mean counter value per launch."""
    per_dispatch = defaultdict(float)  # a dispatch may have one row per counter instance
    for f in Path(folder).rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                per_dispatch[(short(row["Kernel_Name"]), row["Dispatch_Id"])] += float(row["Counter_Value"])
    tot, cnt = defaultdict(float), defaultdict(int)
    for (k, _), v in per_dispatch.items():
        tot[k] += v
        cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    workload, out_dir, round_dir = sys.argv[1], Path(sys.argv[2]), Path(sys.argv[3])
    round_dir.mkdir(parents=True, exist_ok=True)
    stats = sorted((out_dir / "stats").rglob("*kernel_stats.csv"))
    if stats:
        rows = list(csv.DictReader(open(stats[0], newline="")))
        with open(round_dir / f"{workload}_kernel_stats.csv", "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"]])
    fetch, write = counter_means(out_dir / "fetch"), counter_means(out_dir / "write")
    if not fetch:
        return
    kernels = {k: {"FETCH_SIZE": fetch.get(k, 0.0), "WRITE_SIZE": write.get(k, 0.0)} for k in fetch if k.endswith("_kernel")}
    dom = DOMINANT[workload]
    units, bytes_per_unit = UNITS[workload]
    factor = FETCH_FACTOR.get(dom)
    hbm = None if factor is None else factor * kernels[dom]["FETCH_SIZE"] * 1024.0 + kernels[dom]["WRITE_SIZE"] * 1024.0
    doc = {
        "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --workload %s --steps 3 --warmup 1 "
                   "--no-cpu-baseline (one counter per run)" % workload,
        "note": "FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the bytes of a 16 B/lane "
                "coalesced streaming read, so read bytes = 2 * FETCH_SIZE * 1024",
        "kernels": kernels,
        dom + "_hbm_bytes_per_launch": hbm,
        "algorithmic_bytes_per_launch": units * bytes_per_unit,
        "fetch_size_factor": factor,
        "calibration": None if hbm is None else {
            "what": "the kernel reads every input byte exactly once with 16 B/lane loads, so corrected traffic / "
                    "algorithmic bytes should be 1; a factor of 1 would give about 0.5, a wrong request mix something else",
            "traffic_over_algorithmic": hbm / (units * bytes_per_unit)},
    }
    if workload == "c5":
        # the clouds' passes: points and codes are read with 16 B/lane loads as well (same correction)
        cloud = [k for k in kernels if k.startswith("cloud_")]
        cloud_bytes = 338757828   # bench.py's "cloud_bytes_rank0" for this workload on one GPU
        rd = sum(2.0 * kernels[k]["FETCH_SIZE"] * 1024.0 for k in cloud)
        wr = sum(kernels[k]["WRITE_SIZE"] * 1024.0 for k in cloud)
        doc["cloud_passes"] = {"kernels": cloud, "cloud_bytes": cloud_bytes, "read_bytes": rd, "written_bytes": wr,
                               "read_over_cloud_bytes": rd / cloud_bytes}
    (round_dir / f"{workload}_pmc_hbm.json").write_text(json.dumps(doc, indent=1))
    tfile = ROOT / "profiles" / "traffic.json"
    traffic = json.loads(tfile.read_text()) if tfile.exists() else {}
    traffic[workload] = {"hbm_bytes_per_launch": hbm, "source": f"{round_dir}/{workload}_pmc_hbm.json", "units_per_launch": units}
    tfile.write_text(json.dumps(traffic, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
