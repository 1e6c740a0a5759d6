"""The parity report's end-effector solve (roll / pitch / yaw weighted) from the example's pose -- on atan2's +-pi cut -- and from a pose a few hundredths of a radian off it."""
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, "parallel-ddp_amd")
import numpy as np, pyddp
import test_fp32_bar as t
from oracle_binding import Oracle, default_cfg
kw = {**t.EE_KW, **t.EE_RPY}
for off in (False, True):
    x0, u0, xg = t.ee_start(64, np.float32, off)
    o32, o64 = Oracle(default_cfg(4, **kw), np.float32), Oracle(default_cfg(4, **kw), np.float64)
    r32, r64 = o32.run_ilqr_gpusem(x0, u0, xg), o64.run_ilqr_gpusem(x0.astype(np.float64), u0.astype(np.float64), xg.astype(np.float64))
    s = pyddp.Solver(pyddp.default_config(4, dtype=0, **kw))
    out = s.solve(x0, u0, xg)
    pos = s.plant_eval(9, x0[:14], np.zeros(7, np.float32))[0][:6]
    print("pose %s: tool roll pitch yaw kernel %s | oracle32 %s" % ("off the cut" if off else "the example's", np.round(pos[3:], 6), np.round(o32.ee_pos(x0[:14])[0][3:], 6)))
    print("   alpha kernel %s oracle32 %s oracle64 %s" % (list(out["alphaOut"][0][:6]), list(r32["alphaOut"][:6]), list(r64["alphaOut"][:6])))
    print("   J[1] kernel %.5f oracle32 %.5f oracle64 %.5f   J[2] %.5f %.5f %.5f" % (out["Jout"][0][1], r32["Jout"][1], r64["Jout"][1], out["Jout"][0][2], r32["Jout"][2], r64["Jout"][2]))
