#!/usr/bin/env python3
"""Golden vectors for the KUKA iiwa14 plant from data the REFERENCE HOLDS -- container-only generator.

Inputs (read from /root/reference at generation time, never copied):
  plants/iiwa14.urdf                       link masses / centres of mass / inertias, joint origins, damping 0.5
  examples/WAFR_iLQR_examples.cu:80-98     the start pose and the two gravity-balancing torque vectors u0 the example holds
                                           (one per USE_WAFR_URDF branch): known answers of the bias torque at that pose
  test/printDyn.cu:39-77                   the axis-aligned probe states (deltas x unit vectors) and the "balancing" pose

What it emits (tests/golden/iiwa14_urdf_dynamics.json): inputs AND expected outputs, data only --
  * urdf.link_inertia[7][36], urdf.joint_frame[7][16]  the URDF re-expressed in the layout of the model tables
    (6x6 spatial inertia about the joint frame, [angular; linear], column-major; 4x4 fixed joint frame, column-major);
  * cases[]: state x, control u, and for BOTH model variants (0 = this URDF + the EE_TYPE 1 flange modifiers of
    plants/dynamics_arm.cuh:48-65,338-347; 1 = the "WAFR" constants, which exist only as numbers in initI/initT and are
    taken from the committed table) the mass matrix M, bias torque C (gravity + velocity products), qdd, and the
    Jacobian dqdd/d[q,qd,u] by central differences of THIS implementation;
  * balancing: the example's start pose with the bias torque of both variants next to the u0 vectors the example holds.

The dynamics here are an INDEPENDENT implementation: Featherstone's body-coordinate RNEA + CRBA with Pluecker
transforms (link frames, parent->child recursion), dense solve -- not the reference's world-frame composite-inertia
scheme with an unpivoted Gauss-Jordan that oracle/ restates.  Agreement of the two to 1e-9 pins the oracle's plant
functions; tests/test_urdf_pins.py does the comparison and also checks the committed model tables against `urdf`.

usage: python tests/golden/make_iiwa14_from_urdf.py [/root/reference]
"""
import json
import os
import re
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], float)


def rpy_to_R(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx                       # URDF fixed-axis roll, pitch, yaw


def parse_urdf(path):
    root = ET.parse(path).getroot()
    links = {l.get("name"): l for l in root.findall("link")}
    joints = {j.get("name"): j for j in root.findall("joint")}
    I6, F, damping = [], [], []
    for i in range(1, 8):
        j = joints["iiwa_joint_%d" % i]
        assert j.get("type") == "revolute" and j.find("axis").get("xyz").split() == ["0", "0", "1"]
        assert j.find("child").get("link") == "iiwa_link_%d" % i and j.find("parent").get("link") == "iiwa_link_%d" % (i - 1)
        o = j.find("origin")
        rpy = [float(v) for v in o.get("rpy").split()]
        xyz = [float(v) for v in o.get("xyz").split()]
        Fi = np.eye(4)
        Fi[:3, :3] = rpy_to_R(*rpy)
        Fi[:3, 3] = xyz
        F.append(Fi)
        damping.append(float(j.find("dynamics").get("damping")))
        inert = links["iiwa_link_%d" % i].find("inertial")
        assert [float(v) for v in inert.find("origin").get("rpy").split()] == [0, 0, 0]
        c = np.array([float(v) for v in inert.find("origin").get("xyz").split()])
        m = float(inert.find("mass").get("value"))
        t = inert.find("inertia")
        Ic = np.array([[float(t.get("ixx")), float(t.get("ixy")), float(t.get("ixz"))],
                       [float(t.get("ixy")), float(t.get("iyy")), float(t.get("iyz"))],
                       [float(t.get("ixz")), float(t.get("iyz")), float(t.get("izz"))]])
        cx = skew(c)
        S = np.zeros((6, 6))
        S[:3, :3] = Ic + m * cx @ cx.T
        S[:3, 3:] = m * cx
        S[3:, :3] = m * cx.T
        S[3:, 3:] = m * np.eye(3)
        I6.append(S)
    return I6, F, damping


def committed_tables():
    """IIWA14_SPATIAL_INERTIA / IIWA14_JOINT_FRAME of oracle/iiwa14_model_data.h as numpy arrays [variant][link]."""
    txt = open(os.path.join(ROOT, "oracle", "iiwa14_model_data.h")).read()
    out = {}
    for name, n in (("IIWA14_SPATIAL_INERTIA", 36), ("IIWA14_JOINT_FRAME", 16)):
        body = txt.split(name + "[2][7][%d] = {" % n)[1].split("};")[0]
        body = re.sub(r"/\*.*?\*/", "", body)
        vals = [float(v) for v in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", body)]
        out[name] = np.array(vals).reshape(2, 7, n)
    return out


# ------------------------------------------------------------------ Featherstone, body coordinates
def model_from_tables(I36, F16):
    """(spatial inertia 6x6, fixed frame 4x4) per link from column-major tables."""
    return [np.array(I36[i]).reshape(6, 6).T for i in range(7)], [np.array(F16[i]).reshape(4, 4).T for i in range(7)]


def plux(E, r):
    """Pluecker motion transform for a frame rotated by E (parent->child coordinates) and displaced by r."""
    X = np.zeros((6, 6))
    X[:3, :3] = E
    X[3:, 3:] = E
    X[3:, :3] = -E @ skew(r)
    return X


def crm(v):
    X = np.zeros((6, 6))
    X[:3, :3] = skew(v[:3]); X[3:, 3:] = skew(v[:3]); X[3:, :3] = skew(v[3:])
    return X


S_AXIS = np.array([0, 0, 1.0, 0, 0, 0])


def link_transforms(F, q):
    Xs = []
    for i in range(7):
        c, s = np.cos(q[i]), np.sin(q[i])
        Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        Rt = F[i][:3, :3] @ Rz               # child axes in parent coordinates
        Xs.append(plux(Rt.T, F[i][:3, 3]))
    return Xs


def rnea(I, F, q, qd, qdd, grav):
    Xs = link_transforms(F, q)
    v = [None] * 7; a = [None] * 7; f = [None] * 7
    a0 = np.array([0, 0, 0, 0, 0, grav])    # base accelerating upwards == gravity pulling down
    for i in range(7):
        vp = v[i - 1] if i else np.zeros(6)
        ap = a[i - 1] if i else a0
        v[i] = Xs[i] @ vp + S_AXIS * qd[i]
        a[i] = Xs[i] @ ap + S_AXIS * qdd[i] + crm(v[i]) @ (S_AXIS * qd[i])
        f[i] = I[i] @ a[i] - crm(v[i]).T @ (I[i] @ v[i])
    tau = np.zeros(7)
    for i in range(6, -1, -1):
        tau[i] = S_AXIS @ f[i]
        if i:
            f[i - 1] = f[i - 1] + Xs[i].T @ f[i]
    return tau


def crba(I, F, q):
    Xs = link_transforms(F, q)
    Ic = [m.copy() for m in I]
    for i in range(6, 0, -1):
        Ic[i - 1] += Xs[i].T @ Ic[i] @ Xs[i]
    M = np.zeros((7, 7))
    for i in range(7):
        Fv = Ic[i] @ S_AXIS
        M[i, i] = S_AXIS @ Fv
        j = i
        while j > 0:
            Fv = Xs[j].T @ Fv
            j -= 1
            M[i, j] = M[j, i] = S_AXIS @ Fv
    return M


def forward_dynamics(I, F, x, u, grav, damping=0.5):
    q, qd = x[:7], x[7:]
    M = crba(I, F, q)
    C = rnea(I, F, q, qd, np.zeros(7), grav)
    qdd = np.linalg.solve(M, u - C - damping * qd)
    return qdd, M, C


def fd_jacobian(I, F, x, u, grav, eps=1e-6):
    z = np.concatenate([x, u])
    J = np.zeros((7, 21))
    for j in range(21):
        zp, zm = z.copy(), z.copy()
        zp[j] += eps; zm[j] -= eps
        J[:, j] = (forward_dynamics(I, F, zp[:14], zp[14:], grav)[0] - forward_dynamics(I, F, zm[:14], zm[14:], grav)[0]) / (2 * eps)
    return J


def main():
    I_urdf, F_urdf, damping = parse_urdf(os.path.join(REF, "plants", "iiwa14.urdf"))
    assert damping == [0.5] * 7
    tabs = committed_tables()
    # variant 0 = this URDF with the flange (EE_TYPE 1) modifiers on link 7: mass + WEIGHT_MODIFIER 0.03, inertia and first moment
    # x INERTIA_MODIFIER 3 of the values initI spells out (0.0055, 0.0055, 0.005, 0.024) -- plants/dynamics_arm.cuh:48-65,338-347.
    # Links 1..6 are the URDF as it is.  Link 7's unmodified numbers are the URDF's rounded to 2 digits (0.00548 -> 0.0055).
    I0 = [m.copy() for m in I_urdf]
    l7 = np.zeros((6, 6))
    l7[0, 0] = l7[1, 1] = 0.0055 * 3; l7[2, 2] = 0.005 * 3
    h = 0.024 * 3                                                # m * c_z of the URDF (1.2 * 0.02) times the modifier
    l7[:3, 3:] = skew([0, 0, h]); l7[3:, :3] = skew([0, 0, h]).T
    l7[3:, 3:] = (1.2 + 0.03) * np.eye(3)
    I0[6] = l7
    l7e0 = l7.copy(); l7e0[:3, :] /= 3; l7e0[3:, :3] /= 3; l7e0[3:, 3:] = 1.2 * np.eye(3)     # EE_TYPE 0: modifiers 1 and 0
    I_ee0 = [m.copy() for m in I_urdf]; I_ee0[6] = l7e0
    variants = {0: (I0, F_urdf)}
    variants[1] = model_from_tables(tabs["IIWA14_SPATIAL_INERTIA"][1], tabs["IIWA14_JOINT_FRAME"][1])

    rng = np.random.default_rng(20260930)
    cases = []
    deltas = [-0.66667, -0.33333, 0.0, 0.5, 1.0]                 # test/printDyn.cu:39
    for d in deltas:
        for i in range(14):
            x = np.zeros(14); x[i] = d
            cases.append(dict(name=f"printDyn_delta{d}_axis{i}", x=x, u=np.zeros(7)))
    bal = np.array([-1.5708, 0.7854, 0.5246, -0.5246, 0.3927, 0.5246, 1.5708] + [0.0] * 7)   # test/printDyn.cu:64-77
    cases.append(dict(name="printDyn_balancing_pose", x=bal, u=np.zeros(7)))
    for k in range(12):                                          # the reference's own test distribution, test/testDynGrad.cu:12-19
        cases.append(dict(name=f"testDynGrad_distribution_{k}", x=np.concatenate([rng.normal(0, 2, 7), rng.normal(0, 5, 7)]), u=rng.normal(0, 50, 7)))
    PI = 3.14159
    start = np.array([-0.5 * PI, 0.25 * PI, 0.167 * PI, -0.167 * PI, 0.125 * PI, 0.167 * PI, 0.5 * PI] + [0.0] * 7)   # WAFR_iLQR_examples.cu:80-81

    out_cases = []
    for c in cases:
        rec = dict(name=c["name"], x=c["x"].tolist(), u=c["u"].tolist(), variants={})
        for v, (I, F) in variants.items():
            for grav in (9.81, 0.0):
                qdd, M, C = forward_dynamics(I, F, c["x"], c["u"], grav)
                J = fd_jacobian(I, F, c["x"], c["u"], grav)
                rec["variants"][f"{v}_g{grav}"] = dict(qdd=qdd.tolist(), M=M.ravel().tolist(), C=C.tolist(), dqdd_fd=J.T.ravel().tolist())
        out_cases.append(rec)
    balancing = dict(x=start.tolist(), held_u0={"0": [-0.0000000001, -62.282937, 4.172921, 21.513797, -0.088674, -0.890626, 0.0000000001],
                                                "1": [0.0, -102.9832, 11.1968, 47.0724, 2.5993, -7.0290, -0.0907]},   # WAFR_iLQR_examples.cu:93-98
                     bias={str(v): forward_dynamics(I, F, start, np.zeros(7), 9.81)[2].tolist() for v, (I, F) in variants.items()})
    # the default-URDF u0 the example holds was produced WITHOUT the flange (EE_TYPE 0): this implementation reproduces all its printed digits
    balancing["bias"]["0_ee_type0"] = forward_dynamics(I_ee0, F_urdf, start, np.zeros(7), 9.81)[2].tolist()
    doc = dict(_provenance="generated by tests/golden/make_iiwa14_from_urdf.py from /root/reference/plants/iiwa14.urdf, "
                           "examples/WAFR_iLQR_examples.cu:80-98, test/printDyn.cu:39-77; independent body-coordinate RNEA/CRBA (numpy, float64)",
               layout=dict(M="row-major 7x7", dqdd_fd="[col*7+row], cols = dq(7), dqd(7), du(7) -- the plug-in layout (utils/integrators.cuh:17)",
                           link_inertia="column-major 6x6 about the joint frame, [angular; linear]", joint_frame="column-major 4x4"),
               urdf=dict(link_inertia=[m.T.ravel().tolist() for m in I_urdf], joint_frame=[f.T.ravel().tolist() for f in F_urdf], damping=damping,
                         link7_flange=dict(INERTIA_MODIFIER=3, WEIGHT_MODIFIER=0.03, cite="plants/dynamics_arm.cuh:48-65,338-347")),
               balancing=balancing, cases=out_cases)
    path = os.path.join(HERE, "iiwa14_urdf_dynamics.json")
    json.dump(doc, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes;", len(out_cases), "cases")
    for v, h in (("0_ee_type0", "0"), ("0", "0"), ("1", "1")):
        print("variant", v, "bias at the example's start pose", np.round(balancing["bias"][v], 6), "held u0", balancing["held_u0"][h])
    assert np.abs(np.array(balancing["bias"]["0_ee_type0"]) - np.array(balancing["held_u0"]["0"])).max() < 1e-6


if __name__ == "__main__":
    main()
