#!/usr/bin/env python3
"""Summarises a rocprofv3 run (rocpd sqlite .db or kernel-trace CSV) into per-kernel totals:
calls, total ms, average us, share of GPU kernel time. Usage: prof_summary.py <results.db|kernel_trace.csv> [out.md]"""
import csv
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")


def from_db(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels").fetchall()
    return [(n, e - s) for n, s, e in rows]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    path = sys.argv[1]
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    agg = defaultdict(lambda: [0, 0])
    for n, d in rows:
        a = agg[short(n)]
        a[0] += 1
        a[1] += d
    total = sum(v[1] for v in agg.values()) or 1
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {n} | {c} | {d / 1e6:.3f} | {d / c / 1e3:.2f} | {100 * d / total:.1f} |")
    lines.append(f"| TOTAL | {sum(v[0] for v in agg.values())} | {total / 1e6:.3f} | | 100 |")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(text + "\n")


if __name__ == "__main__":
    main()
