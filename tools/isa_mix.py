#!/usr/bin/env python3
"""Static instruction mix per kernel from a hipcc -save-temps .s file. Usage: isa_mix.py file.s [kernel-substring ...]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
want = sys.argv[2:]
# split into functions: label line "name:" followed by code until ".Lfunc_end"
for m in re.finditer(r"^(_Z[\w]+):[^\n]*\n(.*?)^\.Lfunc_end", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want and not any(w in name for w in want):
        continue
    ins = []
    for l in body.splitlines():
        l = l.strip()
        if not l or l.startswith((".", ";", "//")) or l.endswith(":"):
            continue
        ins.append(l.split()[0])
    c = Counter(ins)
    g = Counter()
    for i, n in c.items():
        if "f64" in i: g["f64"] += n
        elif i.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64")): g["int_mul"] += n
        elif i.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")): g["div_seq"] += n
        elif i.startswith(("v_rcp", "v_sqrt", "v_rsq")): g["rcp_sqrt"] += n
        elif i.startswith("ds_"): g["lds"] += n
        elif i.startswith(("global_", "flat_", "buffer_")): g["vmem"] += n
        elif i.startswith("scratch_"): g["scratch"] += n
        elif i.startswith("s_waitcnt"): g["waitcnt"] += n
        elif i.startswith(("s_cbranch", "s_branch")): g["branch"] += n
        elif i.startswith("s_"): g["salu"] += n
        elif i.startswith("v_cndmask"): g["cndmask"] += n
        elif i.startswith("v_"): g["valu"] += n
        else: g["other"] += n
    print(f"{name[:70]:70s} n={len(ins):6d} " + " ".join(f"{k}={v}" for k, v in sorted(g.items())))
