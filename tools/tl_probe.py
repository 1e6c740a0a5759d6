#!/usr/bin/env python3
"""Resources + static instruction mix of the two thread-lane kernels bench.py times (k_fp_tl<float,1,false,false>, k_nis_tl<float,1,false,true>) from a REDUCED copy of
csrc/pddp_tl.hip (the launchers' explicit instantiations removed, the two kernels instantiated explicitly): compiles in ~15 s instead of ~90.
usage: tools/tl_probe.py [-D... | other compiler flags]      several flag sets separated by '--' run in parallel"""
import collections, os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mx_isa
pkg = mx_isa.pkg
src = open(os.path.join(pkg, "csrc", "pddp_tl.hip")).read()
src = "\n".join(l for l in src.splitlines() if not l.startswith("template void launch_"))
src += """
namespace pddp {
template __global__ void k_fp_tl<float, 1, false, false>(Buffers<float>, Dims, CostWeights<float>, float, float, int);
template __global__ void k_nis_tl<float, 1, false, true>(Buffers<float>, Dims, CostWeights<float>, float, float, int, int);
}
"""
os.makedirs("/tmp/isa", exist_ok=True)
open(os.path.join(pkg, "csrc", "_tl_probe.hip"), "w").write(src)


def one(idx_flags):
    idx, flags = idx_flags
    out = f"/tmp/isa/tlp{idx}.s"
    asm, rem = mx_isa.compile_asm("csrc/_tl_probe.hip", flags, out)
    res = mx_isa.resources(rem)
    lines = []
    for nm, body in mx_isa.kernel_bodies(asm).items():
        if not re.search(r"k_fp_tlIfLi1ELb0ELb0|k_nis_tlIfLi1ELb0ELb1", nm):
            continue
        c = collections.Counter()
        for l in body:
            t = l.strip()
            if not l.startswith("\t") or not t or t[0] in ".;":
                continue
            op = t.split()[0]
            c["valu" if op.startswith("v_") else "scratch" if op.startswith("scratch_") else "ds" if op.startswith("ds_") else "other"] += 1
        lines.append(f"  {nm[8:16]:10s} VALU {c['valu']:5d} scratch-ops {c['scratch']:3d} ds {c['ds']:3d} {res.get(nm)}")
    return " ".join(flags) + "\n" + "\n".join(lines)


sets, cur = [], []
for a in sys.argv[1:]:
    if a == "--":
        sets.append(cur); cur = []
    else:
        cur.append(a)
sets.append(cur)
try:
    with ThreadPoolExecutor(max_workers=6) as ex:
        for r in ex.map(one, enumerate(sets)):
            print(r, flush=True)
finally:
    os.remove(os.path.join(pkg, "csrc", "_tl_probe.hip"))
