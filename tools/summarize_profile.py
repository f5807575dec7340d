#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small, tracked summaries under profiles/.

  python tools/summarize_profile.py ROUND TRACE_DIR PMC_DIR [MFMA_DIR]

  TRACE_DIR : output of  rocprofv3 --kernel-trace --stats --output-format csv -d TRACE_DIR -- python bench.py ...
  PMC_DIR   : output of  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d PMC_DIR -- python bench.py ...
  MFMA_DIR  : output of  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d MFMA_DIR -- ...

Writes profiles/<ROUND>_kernel_stats.csv (the rocprofv3 --stats table, kernel names shortened),
       profiles/<ROUND>_pmc_fetch_size.csv (per kernel+grid: launches, mean FETCH_SIZE, corrected HBM bytes),
       profiles/<ROUND>_pmc_fetch.json (kernel short-name -> corrected bytes per launch; bench.py reads it for `traffic`),
       profiles/<ROUND>_pmc_mfma.csv (matrix-core utilisation of the prefill GEMM: SQ_VALU_MFMA_BUSY_CYCLES summed over the
       chip's 1024 SIMDs / (kernel duration x 2.4 GHz x 1024 SIMDs); the counter ticks 32 cycles per 32x32x16 bf16 MFMA).

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is reported in KiB and counts 128-byte
requests as 64 bytes for wide coalesced streaming reads => HBM bytes = FETCH_SIZE * 1024 * 2.  Calibration in
this repo: the read-out GEMV streams 135.0 MB of codes+scales+biases per launch; 2 * FETCH_SIZE KiB = 135.4 MB.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name: str) -> str:
    name = name.strip('"')
    name = re.sub(r"^void ", "", name)
    name = name.replace("uzu::k::", "").replace("uzu::", "")
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^(]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    rnd, trace_dir, pmc_dir = sys.argv[1:4]
    mfma_dir = sys.argv[4] if len(sys.argv) > 4 else None
    os.makedirs("profiles", exist_ok=True)
    newest = lambda pattern, d: sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True), key=os.path.getmtime)[-1:]  # gpurun merges, never deletes
    stats = newest("*_kernel_stats.csv", trace_dir)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(f"profiles/{rnd}_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_us", "avg_us", "pct", "min_us", "max_us"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], f"{int(r['TotalDurationNs']) / 1e3:.1f}", f"{float(r['AverageNs']) / 1e3:.3f}",
                            r["Percentage"], f"{int(r['MinNs']) / 1e3:.3f}", f"{int(r['MaxNs']) / 1e3:.3f}"])
    pmc = newest("*_counter_collection.csv", pmc_dir)
    if pmc:
        agg = collections.defaultdict(lambda: [0, 0.0, 0])
        for r in csv.DictReader(open(pmc[0])):
            if r["Counter_Name"] != "FETCH_SIZE":
                continue
            a = agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        out = {}
        with open(f"profiles/{rnd}_pmc_fetch_size.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "grid_threads", "launches", "mean_FETCH_SIZE_KiB", "hbm_bytes_per_launch(=KiB*1024*2)", "mean_duration_us"])
            for (k, grid), (n, total, dur) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                hbm = total / n * 1024 * 2
                w.writerow([k, grid, n, f"{total / n:.2f}", int(hbm), f"{dur / n / 1e3:.2f}"])
                out[f"{k}|{grid}"] = int(hbm)
        json.dump(out, open(f"profiles/{rnd}_pmc_fetch.json", "w"), indent=1, sort_keys=True)
    mf = newest("*_counter_collection.csv", mfma_dir) if mfma_dir else []
    if mf:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(mf[0])):
            a = agg[short(r["Kernel_Name"])]
            a[r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES":
                a["_n"] += 1
                a["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        with open(f"profiles/{rnd}_pmc_mfma.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "launches", "mean_duration_us", "mean_SQ_VALU_MFMA_BUSY_CYCLES", "mean_SQ_BUSY_CYCLES", "mean_SQ_WAVES",
                        "mfma_util(=MFMA_BUSY/(dur*2.4GHz*1024 SIMDs))"])
            for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
                n = max(a["_n"], 1.0)
                dur_ns = a["_ns"] / n
                busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
                if busy <= 0:
                    continue
                w.writerow([k, int(n), f"{dur_ns / 1e3:.2f}", int(busy), int(a.get("SQ_BUSY_CYCLES", 0.0) / n), int(a.get("SQ_WAVES", 0.0) / n),
                            f"{busy / (dur_ns * 2.4 * 1024):.4f}"])


if __name__ == "__main__":
    main()
