#!/usr/bin/env python3
"""Golden vectors for the pendulum / cart-pole / quadrotor plug-ins -- container-only generator, emits DATA only.

The three plants are straight-line arithmetic (plants/dynamics_pend.cuh:30-51, dynamics_cart.cuh:30-76, dynamics_quad.cuh:42-169) that
does not compile against the v1 helpers and cannot be built here at all (CUDA headers).  This script reads
the bodies of `dynamics` and `dynamicsGradient` from the reference files AT GENERATION TIME, rewrites each C statement mechanically into
a Python assignment (drop the `T ` declarators and the thread-partition scaffolding, `s_xk/s_x -> x`, `s_uk/s_u -> u`, `s_qddk -> qdd`,
`#define`s -> constants) and executes them in float64 on stored inputs.  The outputs are therefore the reference's OWN formulas evaluated
in double precision -- inputs and expected outputs go to tests/golden/closed_form_plants.json; no reference text is stored.

It also tabulates the diagonal cost weights the reference's cost_{pend,cart,quad}.cuh select for the BASELINE horizons (QR(i), R, QF) and
the hover thrust the example holds for the quadrotor (examples/WAFR_iLQR_examples.cu:90: u = 1.22625 = m g / 4).

usage: python tests/golden/make_closed_form_plants.py [/root/reference]
"""
import json
import math
import os
import re
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
DIMS = {"pend": (1, 2, 1), "cart": (2, 4, 1), "quad": (6, 12, 4)}     # npos, n, m (config.cuh:24-40)


def macros(txt, extra=None):
    env = dict(extra or {})
    for m in re.finditer(r"^\s*#define\s+(\w+)\s+([^/\n]+?)\s*(?://.*)?$", txt, re.M):
        name, val = m.group(1), m.group(2).strip()
        if "(" in name or not val:
            continue
        try:
            env.setdefault(name, eval(val, {}, env))
        except Exception:
            pass
    return env


def body_of(txt, signature, op="{", cl="}"):
    if signature == "(":
        op, cl = "(", ")"
    i = txt.index(signature)
    j = txt.index(op, i)
    depth, k = 0, j
    while True:
        depth += txt[k] == op
        depth -= txt[k] == cl
        if depth == 0:
            return txt[j + 1:k]
        k += 1


def to_python(body):
    body = re.sub(r"//[^\n]*", "", body)
    body = re.sub(r"#ifdef __CUDA_ARCH__.*?#endif", "", body, flags=re.S)
    body = re.sub(r"for\s*\([^)]*\)", "", body)                   # the thread-partition loop over `reps` (1 here)
    body = re.sub(r"if \(s_qdd != nullptr\)", "", body)
    out = []
    for st in body.replace("{", ";").replace("}", ";").split(";"):
        st = st.strip()
        if not st or st.startswith(("int start", "singleLoopVals", "for(", "for (", "if (s_qdd", "dynamics(", "#pragma", "memset")):
            continue
        if re.match(r"T \*s_\w+k = ", st):                       # T *s_xk = &s_x[STATE_SIZE*iter]  (reps = 1: iter = 0)
            continue
        st = re.sub(r"^T\s+", "", st)
        st = re.sub(r"\bs_xk\b|\bs_x\b", "x", st)
        st = re.sub(r"\bs_uk\b|\bs_u\b", "u", st)
        st = re.sub(r"\bs_qddk\b", "qdd", st)
        st = re.sub(r"\bs_dqdd\b", "dqdd", st)
        out.append(st)
    return "\n".join(out)


def evaluate(plant, x, u):
    npos, n, m = DIMS[plant]
    txt = open(os.path.join(REF, "plants", f"dynamics_{plant}.cuh")).read()
    env = macros(txt, dict(NUM_POS=npos, STATE_SIZE=n, CONTROL_SIZE=m))
    env.update(cos=math.cos, sin=math.sin)
    dyn = to_python(body_of(txt, "void dynamics(T *s_qdd"))
    grad = to_python(body_of(txt, "void dynamicsGradient(T *s_dqdd"))
    sc = dict(env, x=list(x), u=list(u), qdd=[0.0] * npos)
    exec(dyn, {}, sc)
    sg = dict(env, x=list(x), u=list(u), dqdd=[0.0] * (npos * (n + m)))
    exec(grad, {}, sg)
    return [float(v) for v in sc["qdd"]], [float(v) for v in sg["dqdd"]]


def cost_weights(plant, N):
    txt = open(os.path.join(REF, "plants", f"cost_{plant}.cuh")).read()
    if plant == "cart":                                           # weights depend on NUM_TIME_STEPS (cost_cart.cuh:19-37): pick the branch
        blocks = re.split(r"#if NUM_TIME_STEPS == 512|#elif NUM_TIME_STEPS == 256|#else", txt.split("#define QR(i)")[0])
        txt_w = blocks[1] if N == 512 else blocks[2] if N == 256 else blocks[3]
    else:
        txt_w = txt.split("#define QR(i)")[0].split("#endif")[-1]
    env = macros(txt_w)
    qr = re.search(r"#define QR\(i\) (.*)", txt).group(1)
    npos, n, m = DIMS[plant]

    def tern(s, e):                                               # C "c ? a : b" (right-associative, parenthesised), recursively
        s = s.strip()
        while s.startswith("(") and s.endswith(")") and body_of(s, "(") == s[1:-1]:
            s = s[1:-1].strip()
        depth = 0
        for k, ch in enumerate(s):
            depth += ch == "("; depth -= ch == ")"
            if ch == "?" and depth == 0:
                d2 = 0
                for k2 in range(k + 1, len(s)):
                    d2 += s[k2] == "("; d2 -= s[k2] == ")"
                    if s[k2] == ":" and d2 == 0:
                        return tern(s[k + 1:k2], e) if eval(s[:k], {}, e) else tern(s[k2 + 1:], e)
        return float(eval(s, {}, e))

    def QR(i):
        return tern(qr, dict(env, i=i))

    return dict(running=[QR(i) for i in range(n + m)], final=[float(env["QF"])] * n + [0.0] * m)


def main():
    rng = np.random.default_rng(20260930)
    doc = {"_provenance": "generated by tests/golden/make_closed_form_plants.py: the reference's own dynamics / dynamicsGradient statements "
                          "(plants/dynamics_{pend,cart,quad}.cuh) executed in float64 on the stored inputs; cost weights from plants/cost_*.cuh",
           "layout": {"dqdd": "[col*npos + row], cols = dq, dqd, du (utils/integrators.cuh:17)"}}
    for plant, (npos, n, m) in DIMS.items():
        cases = [dict(x=[0.1 * (i + 1) for i in range(n)], u=[0.5 + i for i in range(m)])]          # the survey's probe point
        for _ in range(24):
            cases.append(dict(x=rng.normal(0, 1.0, n).tolist(), u=rng.normal(0, 3.0, m).tolist()))
        if plant == "quad":                                       # the example's start: hover (WAFR_iLQR_examples.cu:28-33,78,90)
            cases.append(dict(x=[0, 0, 0.5] + [0.0] * 9, u=[1.22625] * 4, note="hover: the held u0 balances gravity"))
        for c in cases:
            c["qdd"], c["dqdd"] = evaluate(plant, c["x"], c["u"])
        doc[plant] = dict(cases=cases, cost_weights={str(N): cost_weights(plant, N) for N in (64, 128, 256)})
    assert max(abs(v) for v in doc["quad"]["cases"][-1]["qdd"]) < 1e-12
    path = os.path.join(HERE, "closed_form_plants.json")
    json.dump(doc, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes")
    for p in DIMS:
        print(p, "probe qdd", doc[p]["cases"][0]["qdd"], "weights N=128", doc[p]["cost_weights"]["128"])


if __name__ == "__main__":
    main()
