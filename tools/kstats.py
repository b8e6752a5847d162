#!/usr/bin/env python3
"""Print a rocprofv3 *kernel_stats.csv with readable kernel names (template / argument lists cut).

    python tools/kstats.py gpurun_out/prof [--top 12]
"""
import csv
import glob
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[0-9, a-z]+>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    root = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
    paths = [root] if root.endswith(".csv") else glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)
    for path in paths:
        print("#", path)
        with open(path) as f:
            rows = list(csv.DictReader(f))
        print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        for r in rows[:top]:
            print(f"{short(r['Name']):70s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.2f} {float(r['MinNs']) / 1e3:10.2f}"
                  f" {float(r['MaxNs']) / 1e3:10.2f} {float(r['Percentage']):6.2f}")


if __name__ == "__main__":
    main()
