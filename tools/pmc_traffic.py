#!/usr/bin/env python3
"""Folds rocprofv3 counter-collection CSVs (one `--pmc FETCH_SIZE` pass, one `--pmc WRITE_SIZE` pass and optionally an SQ pass
holding SQ_INSTS_VALU + SQ_BUSY_CYCLES of the same bench command, collected without any tracing) into the per-kernel table
bench.py reads: HBM traffic per launch and the VALU issue utilisation.

usage: tools/pmc_traffic.py <fetch.csv> <write.csv> <batch> <out.json> [sq.csv [valu_mix.json]]

valu_mix.json (tools/valu_mix.py): the kernel's own average issue cost per VALU instruction from its static mix and the measured
per-instruction costs (profiles/rNN_valu_issue.txt) - ~2.3 cycles for plain 32-bit VOP1/VOP2 forms, ~4.1 for packed / three-input /
permute / multiply / f64 forms.  Without it every instruction is priced at 4 cycles (round 2's model: right for the FAST
kernel, whose mix is 90 % 4-cycle forms, 15-20 % too high for resize / descriptor / quadtree)."""
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
        if not name.startswith("k_fast_cells_cols"):       # its two instantiations are two different launches
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
    if len(sys.argv) > 5:
        # SQ_INSTS_VALU wave-instructions x 4 cycles each on one of the 1024 SIMDs, against the kernel's busy cycles
        # (SQ_BUSY_CYCLES is reported summed over the 32 shader engines x ... of the device: normalised as in round 1 by the
        # ratio that makes a pure-VALU micro-kernel read 1.0: busy / 32)
        vi, bc = fold(sys.argv[5], "SQ_INSTS_VALU"), fold(sys.argv[5], "SQ_BUSY_CYCLES")
        mix = json.load(open(sys.argv[6]))["kernels"] if len(sys.argv) > 6 else {}
        for k in kernels:
            if k in vi and k in bc and bc[k][0] > 0:
                inst, busy = vi[k][0] / vi[k][1], bc[k][0] / bc[k][1]
                kernels[k]["SQ_INSTS_VALU_per_launch"] = round(inst)
                kernels[k]["SQ_BUSY_CYCLES_per_launch"] = round(busy)
                cyc = 4.0
                m = mix.get(k) or next((v for n, v in mix.items() if n.split("<")[0] == k), None)
                if m:
                    cyc = m["cycles_per_valu_instruction"]
                    kernels[k]["cycles_per_valu_instruction"] = cyc
                    kernels[k]["valu_share_at_2_cycle_rate"] = m["share_at_2_cycle_rate"]
                # NOT capped: a value above 1 would say the cost model is wrong
                kernels[k]["valu_issue_utilisation"] = round(inst * cyc / 1024.0 / (busy / 32.0), 3)
                kernels[k]["valu_issue_utilisation_flat4"] = round(inst * 4.0 / 1024.0 / (busy / 32.0), 3)
    doc = {
        "_about": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing) over "
                  f"`python3 bench.py --steps 5 --warmup 2 --distinct 8 --no-cpu-baseline --no-extras` (tools/profile_round.sh) (batch {batch}, 640x480), MI355X. Counter "
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
