#!/usr/bin/env python3
"""Static instruction mix + resources of the thread-lane translation unit (csrc/pddp_tl.hip), per kernel.  usage: tools/tl_isa.py [regex] [-D...]"""
import collections, re, sys
import mx_isa
defs = [a for a in sys.argv[1:] if a.startswith("-")]
pat = re.compile(next((a for a in sys.argv[1:] if not a.startswith("-")), "k_fp_tl|k_nis_tl"))
asm, rem = mx_isa.compile_asm("csrc/pddp_tl.hip", defs, "/tmp/isa/tl.s")
res = mx_isa.resources(rem)
for nm, body in mx_isa.kernel_bodies(asm).items():
    if not pat.search(nm):
        continue
    c = collections.Counter()
    for l in body:
        t = l.strip()
        if not l.startswith("\t") or not t or t[0] in ".;":
            continue
        op = t.split()[0]
        k = ("trans" if re.match(r"v_(rcp|sqrt|rsq|sin|cos|exp|log)", op) else "v_mov" if op.startswith("v_mov") else "cndmask" if "cndmask" in op else "valu") if op.startswith("v_") else \
            "ds" if op.startswith("ds_") else "vmem" if re.match(r"(buffer|global|flat)_", op) else "scratch" if op.startswith("scratch_") else "wait" if "waitcnt" in op else "nop" if op == "s_nop" else "salu"
        c[k] += 1
    v = c["valu"] + c["v_mov"] + c["cndmask"] + c["trans"]
    print(f"{nm[:80]:80s} VALU {v:6d} {dict(c)} {res.get(nm)}")
