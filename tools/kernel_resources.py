#!/usr/bin/env python3
"""Prints VGPR / SGPR / scratch / occupancy / LDS of the gfx950 kernels of libpddp (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py [regex on the demangled-ish name]"""
import os, re, subprocess, sys
pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parallel-ddp_amd")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-unused-result", "-Wno-unused-value",
       "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", "csrc/pddp_plant_" + (sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("pend", "cart", "quad", "arm") else "arm") + ".hip"]
out = subprocess.run(cmd, cwd=pkg, capture_output=True, text=True).stderr
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
cur = {}
for ln in out.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", ln)
    if "Function Name:" in ln:
        cur = {"name": ln.split("Function Name: ")[1].split(" [")[0]}
    for key, tag in (("vgpr", " VGPRs: "), ("agpr", "AGPRs: "), ("sgpr", "TotalSGPRs: "), ("scratch", "ScratchSize [bytes/lane]: "), ("occ", "Occupancy [waves/SIMD]: "), ("lds", "LDS Size [bytes/block]: ")):
        if tag in ln:
            cur[key] = ln.split(tag)[1].split(" ")[0]
            if key == "lds" and pat.search(cur.get("name", "")):
                n = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
                n = re.sub(r"\(.*", "", n).replace("void pddp::", "")
                print(f"{n[:70]:70s} vgpr {cur.get('vgpr','?'):>4s} agpr {cur.get('agpr','?'):>3s} sgpr {cur.get('sgpr','?'):>4s} scratch {cur.get('scratch','?'):>5s} occ {cur.get('occ','?'):>2s} lds {cur.get('lds','?'):>6s}")
