#!/usr/bin/env python3
"""Static instruction mix of gfx950 kernels of libpddp.  usage: tools/isa_stats.py <regex on mangled name> [--dump file]"""
import collections, os, re, subprocess, sys
pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parallel-ddp_amd")
asm = "/tmp/isa/pddp.s"
os.makedirs("/tmp/isa", exist_ok=True)
if "--no-build" not in sys.argv:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-result", "-Wno-unused-value",
                    "--cuda-device-only", "-S", "-o", asm, "csrc/pddp_plant_" + (sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("pend", "cart", "quad", "arm") else "arm") + ".hip"], cwd=pkg, capture_output=True)
pat = re.compile(sys.argv[1])
lines = open(asm).read().splitlines()
starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN4pddp\S*:", l)]
for si, st in enumerate(starts):
    name = lines[st].split(":")[0]
    if not pat.search(name):
        continue
    end = next((i for i in range(st, len(lines)) if lines[i].strip().startswith("s_endpgm")), len(lines))
    body = lines[st:end + 1]
    c = collections.Counter()
    n = 0
    for l in body:
        t = l.strip()
        if not l.startswith("\t") or t.startswith(".") or t.startswith(";") or not t:
            continue
        op = t.split()[0]
        n += 1
        if op.startswith("v_"):
            key = "valu_f64" if "f64" in op else ("valu_dpp" if "dpp" in t else ("v_cndmask" if op.startswith("v_cndmask") else ("v_mov" if op.startswith("v_mov") else "valu")))
        elif op.startswith("ds_"): key = op
        elif op.startswith("s_waitcnt"): key = "s_waitcnt"
        elif op.startswith("scratch_"): key = "scratch_ld" if "load" in op else "scratch_st"
        elif op.startswith(("global_", "buffer_", "flat_")): key = "vmem_ld" if "load" in op else "vmem_st"
        elif op.startswith("s_"): key = "salu"
        else: key = op
        c[key] += 1
    print(f"{name[:60]:60s} total {n}")
    print("   ", dict(c.most_common(16)))
    if "--dump" in sys.argv:
        open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
