#!/usr/bin/env python3
"""ISA of the matrix-core translation unit (csrc/pddp_mx.hip) for gfx950: per-kernel resources, and the body of one kernel to a file.
usage: tools/mx_isa.py [-D...] [--dump <regex on the mangled name> <file>]      (cross-compiles: no GPU needed)"""
import os, re, subprocess, sys
pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parallel-ddp_amd")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-result", "-Wno-unused-value", "-fno-slp-vectorize"]


HIPCC = "/opt/rocm/bin/hipcc"


def compile_asm(src="csrc/pddp_mx.hip", defs=(), out="/tmp/isa/mx.s"):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    r = subprocess.run([HIPCC] + FLAGS + list(defs) + ["--cuda-device-only", "-S", "-Rpass-analysis=kernel-resource-usage", "-o", out, src], cwd=pkg, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("device compile of %s failed:\n%s" % (src, r.stderr[-3000:]))
    return open(out).read(), r.stderr


def kernel_bodies(asm):
    """{mangled name: [lines]} of every kernel (label ... .Lfunc_end)"""
    lines = asm.splitlines()
    res = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN4pddp\S+):\s*(;.*)?$", l)
        if m:
            end = next((j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end")), len(lines) - 1)      # (a kernel may hold several s_endpgm: early exits)
            res[m.group(1)] = [x for x in lines[i + 1:end]]
    return res


def resources(remarks):
    cur, d, out = None, {}, {}
    for ln in remarks.splitlines():
        if "Function Name:" in ln:
            cur, d = ln.split("Function Name: ")[1].split(" [")[0], {}
        for k, tag in (("vgpr", " VGPRs: "), ("scratch", "ScratchSize [bytes/lane]: "), ("occ", "Occupancy [waves/SIMD]: "), ("lds", "LDS Size [bytes/block]: ")):
            if tag in ln:
                d[k] = int(ln.split(tag)[1].split(" ")[0])
                if k == "lds":
                    out[cur] = d
    return out


if __name__ == "__main__":
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    asm, rem = compile_asm(defs=defs)
    for nm, d in resources(rem).items():
        print(f"{nm[:72]:72s} {d}")
    if "--dump" in sys.argv:
        i = sys.argv.index("--dump")
        pat = re.compile(sys.argv[i + 1])
        for nm, body in kernel_bodies(asm).items():
            if pat.search(nm):
                open(sys.argv[i + 2], "w").write("\n".join(body))
                print("dumped", nm, len(body), "lines")
                break


# ---- the hand-counted wait of the matrix-core backward pass's operand prefetch (csrc/bp_mfma.hpp: kMxGainStores, kMxCtgStores) ---------------------------------------
K_GAIN, K_CTG = 4, 3


def prefetch_loop_ops(body):
    """The memory-side instruction sequence of the knot loop of one k_bp_mfma instantiation, in program order:
    ("dma" | "load" | "gain" | "store" | "wait:<n>" | "cwait:<n>", text).  dma = buffer_load ... lds; gain = a buffer store of any width; store = any other store;
    wait = an s_waitcnt vmcnt(n) of the source's inline assembly, cwait = one the compiler placed; the loop = the cycle around the "Inner Loop Header" label."""
    hdr = next((i for i, l in enumerate(body) if "Inner Loop Header" in l), None)
    if hdr is None:
        return None
    label = body[hdr].split(":")[0].strip()
    # a block right in front of the header that every back edge may also target (the compiler hoists the all-slots wait there)
    pre = next((i for i in range(hdr - 1, max(hdr - 12, 0), -1) if re.match(r"^\.LBB\d+_\d+:", body[i])), None)
    pre_label = body[pre].split(":")[0].strip() if pre is not None else None
    def targets(l, lab):
        return lab is not None and re.search(r"s_c?branch\S*\s+" + re.escape(lab) + r"\b", l)
    back = [i for i in range(hdr, len(body)) if targets(body[i], label) or targets(body[i], pre_label)]
    if not back:
        return None
    start = pre if (pre_label and any(targets(body[i], pre_label) for i in back)) else hdr
    ops, in_asm = [], False
    for l in body[start:back[-1] + 1]:
        t = l.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True
        elif t.startswith(";;#ASMEND"): in_asm = False
        if not l.startswith("\t") or not t or t[0] in ".;":
            continue
        op = t.split()[0]
        m = re.search(r"vmcnt\((\d+)\)", t)
        if op == "s_waitcnt" and m:
            ops.append((("wait:" if in_asm else "cwait:") + m.group(1), t))
        elif op.startswith("buffer_load") and t.endswith("lds"):
            ops.append(("dma", t))
        elif re.match(r"(buffer|global|flat|scratch)_load", op):
            ops.append(("load", t))
        elif op.startswith("buffer_store_dword"):
            ops.append(("gain", t))                                       # (any width: the gain stores and the cost-to-go stores are told apart by their position)
        elif re.match(r"(buffer|global|flat|scratch)_store", op):
            ops.append(("store", t))
    return ops


def check_prefetch_invariant(body, hqq=False, exact=True):
    """Violations (strings) of what `s_waitcnt vmcnt(kMxGainStores [+ kMxCtgStores])` at the top of a knot relies on; [] = the emitted loop is what the source counts on.
    Stores in program order behind the prefetch group: K_GAIN gain stores (K | du), then K_CTG cost-to-go stores ([P | p]) -- buffer stores of any width."""
    ops = prefetch_loop_ops(body)
    if ops is None:
        return ["no knot loop found"]
    kinds = [k for k, _ in ops]
    bad = []
    waits = sorted(int(k.split(":")[1]) for k in kinds if k.startswith("wait:"))
    if waits != [K_GAIN, K_GAIN + K_CTG]:
        bad.append(f"inline waits of the loop are vmcnt{waits}, expected [{K_GAIN}, {K_GAIN + K_CTG}]")
    mem = [k for k in kinds if not k.startswith("wait:")]
    n_dma = 5 if hqq else 4
    if mem[:n_dma] != ["dma"] * n_dma:
        bad.append(f"the knot's first memory instructions are {mem[:n_dma + 1]}, expected {n_dma} LDS-direct loads right behind the wait")
    rest = mem[n_dma:]
    if "dma" in rest or "load" in rest:
        bad.append("a load behind the prefetch group: it would be counted as one of the stores")
    if any(k.startswith("cwait:") for k in rest) or any(k.startswith("cwait:") for k in mem[:n_dma]):
        bad.append("a compiler-placed vmcnt wait inside the loop: something other than the counted stores is in flight")
    # exact: the instantiation issues nothing but the counted stores (every one but M > 1 without the fused sweep maps, whose A - B K | B du stores come on top: there the
    # wait is conservative -- it also sits out some stores -- but still right, because what the count needs is AT LEAST that many stores behind the prefetch)
    stores = [k for k in rest if k in ("gain", "store")]
    if exact and "store" in stores:
        bad.append("a store that is not a buffer store behind the prefetch group")
    if (len(stores) != K_GAIN + K_CTG) if exact else (len(stores) < K_GAIN + K_CTG):
        bad.append(f"{len(stores)} buffer stores per knot, the waits count on exactly {K_GAIN} gain stores (kMxGainStores) + {K_CTG} cost-to-go stores (kMxCtgStores)")
    return bad
