#!/usr/bin/env python3
"""Turn rocprofv3 PMC csv output into the per-launch HBM traffic record bench.py embeds.

    python tools/parse_pmc.py --fetch <counter_collection.csv> --write <counter_collection.csv> \
        --kernel k_verify --out profiles/r01_traffic.json

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming
read, so the read side is doubled; WRITE_SIZE is exact for streaming stores.  The two counters
come from SEPARATE passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2).
"""
import argparse
import csv
import json


def mean_counter(path, kernel, counter):
    vals = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel} in {path}")
    return sum(vals) / len(vals), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--kernel", default="k_verify")
    ap.add_argument("--algorithmic-bytes", type=int, default=77861504)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    fetch_kib, nf = mean_counter(a.fetch, a.kernel, "FETCH_SIZE")
    write_kib, nw = mean_counter(a.write, a.kernel, "WRITE_SIZE")
    read_bytes = fetch_kib * 1024 * 2          # gfx950: FETCH_SIZE counts 128-B requests as 64 B
    write_bytes = write_kib * 1024
    rec = dict(kernel=a.kernel, launches_fetch_pass=nf, launches_write_pass=nw,
               FETCH_SIZE_KiB_mean=fetch_kib, WRITE_SIZE_KiB_mean=write_kib,
               read_bytes_per_launch=read_bytes, write_bytes_per_launch=write_bytes,
               hbm_bytes_per_launch=read_bytes + write_bytes, algorithmic_bytes=a.algorithmic_bytes,
               traffic_over_algorithmic=(read_bytes + write_bytes) / a.algorithmic_bytes,
               corrections="FETCH_SIZE KiB x1024 x2 (gfx950 half-count of 16 B/lane streaming reads); "
                           "WRITE_SIZE KiB x1024; separate --pmc passes",
               command="rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python3 bench.py "
                       "--steps 40 --warmup 10 --no-cpu-baseline")
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
