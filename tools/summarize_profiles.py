#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/...) into the small, tracked files under profiles/.

    python tools/summarize_profiles.py <round-tag> --stats <dir with *_kernel_stats.csv> \
        [--fetch <dir with FETCH_SIZE counter_collection.csv>] [--write <dir with WRITE_SIZE ...>] [--bench bench.json]

Outputs  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, library kernels only, short names
         profiles/<tag>_pmc_traffic.json   per kernel and grid size: mean FETCH_SIZE / WRITE_SIZE (KiB, raw counters) and
                                           the HBM bytes per launch with the gfx950 correction of MI355X_MICROARCH.md
                                           (FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read: x2)
         profiles/<tag>_bench.json         the bench line produced under the profiler (if given)
"""
import argparse
import collections
import csv
import glob
import json
import re
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def short_name(n: str):
    m = re.search(r"parrot::(\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--bench")
    a = ap.parse_args()
    out = REPO / "profiles"
    out.mkdir(exist_ok=True)
    if a.stats:
        f = glob.glob(f"{a.stats}/**/*_kernel_stats.csv", recursive=True)[0]
        rows = [r for r in csv.DictReader(open(f)) if short_name(r["Name"])]
        with open(out / f"{a.tag}_kernel_stats.csv", "w", newline="") as fo:
            w = csv.writer(fo)
            w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short_name(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
        print("wrote", out / f"{a.tag}_kernel_stats.csv", len(rows), "kernels")
    pmc = collections.defaultdict(dict)
    for counter, d in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
        if not d:
            continue
        f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = short_name(r["Kernel_Name"])
            if k and r["Counter_Name"] == counter:
                agg[f"{k} grid={r['Grid_Size']} wg={r['Workgroup_Size']}"].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pmc[k][counter + "_KiB_mean"] = sum(v) / len(v)
            pmc[k]["launches_" + counter] = len(v)
    for k, v in pmc.items():
        fetch, write = v.get("FETCH_SIZE_KiB_mean"), v.get("WRITE_SIZE_KiB_mean")
        if fetch is not None:
            v["hbm_bytes_per_launch_corrected"] = (2 * fetch + (write or 0.0)) * 1024
    if pmc:
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; hbm bytes = (2*FETCH + WRITE) KiB "
                           "(gfx950: FETCH_SIZE reports half the bytes of a wide coalesced read)", "kernels": dict(sorted(pmc.items()))},
                  open(out / f"{a.tag}_pmc_traffic.json", "w"), indent=1)
        print("wrote", out / f"{a.tag}_pmc_traffic.json")
    if a.bench:
        line = [l for l in open(a.bench) if l.startswith("{")][-1]
        json.dump(json.loads(line), open(out / f"{a.tag}_bench.json", "w"), indent=1)
        print("wrote", out / f"{a.tag}_bench.json")


if __name__ == "__main__":
    main()
