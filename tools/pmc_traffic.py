#!/usr/bin/env python3
"""Folds two rocprofv3 counter-collection CSVs (one `--pmc FETCH_SIZE` pass, one `--pmc WRITE_SIZE` pass of the same
bench command, collected without any tracing) into the per-kernel HBM traffic table bench.py reads.

usage: tools/pmc_traffic.py <fetch.csv> <write.csv> <batch> <out.json>"""
import collections
import csv
import json
import sys


def fold(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        if name.startswith("void "):
            name = name[5:]
        name = name.split("<")[0]
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return acc


def main():
    fetch, write, batch, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, w = fold(fetch, "FETCH_SIZE"), fold(write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        kernels[k] = {
            "FETCH_SIZE_KB_per_launch": round(f[k][0] / max(f[k][1], 1), 1), "launches_sampled_FETCH_SIZE": f[k][1],
            "WRITE_SIZE_KB_per_launch": round(w[k][0] / max(w[k][1], 1), 1), "launches_sampled_WRITE_SIZE": w[k][1],
        }
        kernels[k]["HBM_BYTES_per_launch"] = round((2.0 * kernels[k]["FETCH_SIZE_KB_per_launch"] + kernels[k]["WRITE_SIZE_KB_per_launch"]) * 1024.0)
    doc = {
        "_about": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing) over "
                  f"`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` (batch {batch}, 640x480), MI355X. Counter "
                  "units are KB as rocprofv3 reports them (TCC_EA0 requests x 64 B / 1024). Calibration in this repo's own "
                  "access widths (tools/ubench_copy.hip under the same two passes): a streaming copy of 86 016 KB reports "
                  "FETCH_SIZE 43 018 KB at 4, 8 and 16 bytes per lane alike (the gfx950 half-count of "
                  "MI355X_MICROARCH.md: 128-B requests tallied at 64 B) and WRITE_SIZE 86 016 KB exactly; 704-byte rows "
                  "read 4 B per lane by 64-thread blocks report 0.64 of their bytes (partial lines). So the *_per_launch "
                  "fields are the raw counters and HBM_BYTES_per_launch = 2 x FETCH + WRITE is what bench.py reports as "
                  "`traffic`. Infinity-Cache hits are included in these memory-side counters. Kernels launched several "
                  "times per step with different grids (k_pyr_resize, k_quadtree variants) are averaged over all their "
                  "launches.",
        "batch": batch,
        "kernels": kernels,
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: (v["FETCH_SIZE_KB_per_launch"], v["WRITE_SIZE_KB_per_launch"]) for k, v in kernels.items()}, indent=0))


if __name__ == "__main__":
    main()
