#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSV output (counter_collection.csv files under <dir>/g*/...) into per-kernel
per-dispatch averages. Usage: pmc_summary.py <dir> [out.md]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")


def main():
    d = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))   # kernel -> counter -> [dispatches, sum]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                a = acc[k][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    counters = sorted({c for k in acc for c in acc[k]})
    lines = []
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", [0, 0])[1]):
        if not k.startswith("k_"):
            continue
        row = acc[k]
        n = max(v[0] for v in row.values())
        lines.append(f"### {k}  ({n} dispatches per pass)")
        for c in counters:
            if c in row:
                lines.append(f"- {c}: total {row[c][1]:.6g}, per dispatch {row[c][1] / row[c][0]:.6g}")
        g = lambda c: row[c][1] if c in row else None
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            lines.append(f"- HBM-side bytes per dispatch (FETCH_SIZE, WRITE_SIZE in KiB; raw, no gfx950 x2 read correction): "
                         f"read {g('FETCH_SIZE') * 1024 / row['FETCH_SIZE'][0]:.4g} B, write {g('WRITE_SIZE') * 1024 / row['WRITE_SIZE'][0]:.4g} B")
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
            lines.append(f"- L2 hit rate: {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.4f}")
        if g("SQ_ACTIVE_INST_VALU") is not None and g("SQ_BUSY_CYCLES"):
            lines.append(f"- VALU active / SQ busy cycles: {g('SQ_ACTIVE_INST_VALU') / g('SQ_BUSY_CYCLES'):.4f}")
        if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
            lines.append(f"- mean resident waves (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES): {g('SQ_WAVE_CYCLES') / g('SQ_BUSY_CYCLES'):.3f}")
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            lines.append(f"- active lanes per VALU cycle (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU): "
                         f"{g('SQ_THREAD_CYCLES_VALU') / g('SQ_ACTIVE_INST_VALU'):.2f} of 64")
        if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
            lines.append(f"- VALU instructions per wave: {g('SQ_INSTS_VALU') / g('SQ_WAVES'):.1f}")
        lines.append("")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
