"""Design study (CPU, float64; round 5): what a distance bound taken DURING the walk would save on the
reflection / primary rays of C3 -- without the literal evaluation in the loop.  The bound: a leaf whose filter quantities make the
hit certain (origin outside the sphere, discriminant clear of zero) bounds the nearest hit by its centre's projection T_c = -b;
from then on the walk tests nodes from the point o + (T_c + delta) d looking back (the reversed walk the shadow rays already use),
which drops what lies beyond at no cost per node.  Compared: no culling (the kernel as it stands for these rays), this bound,
and the exact nearest hit so far (tools/bvh_sim.py's TCULL: the upper limit of any such scheme).  Same threaded 4-ary hierarchy
and DFS order as rt_bvh_build.h's (sahdiag / ritter / 4).  usage: python tools/bvh_cull_sim.py [C3] [ntiles=24]"""
import math, os, sys
import numpy as np
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["C3", "24"])
os.environ["ONLY"] = "__none__"          # import the simulator's builders and ray generator without its sweep
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bvh_sim as S

rec, link = S.build(S.split_sah_diag, S.bound_ritter, 4, 4)
C, R, KAPPA = S.C, S.R, S.KAPPA
n = len(rec)

def walk(o, d, mode):
    """mode 0: forward sign-aware test only; 1: conservative bound + reversed test; 2: exact nearest hit so far"""
    i = 0; tests = 0; cands = 0; switches = 0
    best = 1e30; tau = 1e30
    while i < n:
        c, r = rec[i]; kind, x = link[i]
        oc = o - c
        b = oc @ d; cc = oc @ oc - r * r
        bm = min(b, 0.0)
        ok = bm * bm * (1 + KAPPA) ** 2 - cc + (r * r * KAPPA if kind == 'l' else 0.0) > 0
        if ok and mode == 1 and tau < 1e29:
            w = o + (tau + 0.0051 + 1e-4 * tau) * d         # just behind the certain hit's centre projection
            wc = w - c
            b2 = -(wc @ d); c2 = wc @ wc - r * r            # looking back along -d
            ok = min(b2, 0.0) ** 2 * (1 + KAPPA) ** 2 - c2 + (r * r * KAPPA if kind == 'l' else 0.0) + 2.5e-4 * (tau + 0.0051) ** 2 > 0
        if ok and mode == 2 and cc > 0 and best < 1e29:
            ok = (-b - math.sqrt(max(b * b - cc, 0.0))) < best
        tests += 1
        if kind == "l":
            if ok:
                cands += 1
                disc = b * b - cc
                if disc > 0 and b < 0:
                    t = -b - math.sqrt(disc)
                    if t > 1e-3 and t < best: best = t
                    # certain by the filter's own quantities: outside the sphere, in front, discriminant clear of zero
                    if mode == 1 and cc > 1e-3 * r * r and disc > 1e-2 * r * r and -b < tau:
                        tau = -b; switches += 1
            i += 1
        else:
            i = i + 1 if ok else x
    return best, tests, cands, switches

rays = S.gen_rays(*S.build(S.split_median, S.bound_box))
fwd = [(o, d) for k, (t, o, d) in enumerate(rays) if not np.array_equal(o, S.light)]      # primary / reflection rays
print(S.cfgname, "spheres", S.N, "nodes", n, "rays", len(rays), "of them primary / reflection", len(fwd))
res = {}
for mode, name in ((0, "no culling (the kernel today)"), (1, "centre-projection bound + reversed test"), (2, "exact nearest hit so far (limit)")):
    out = [walk(o, d, mode) for (o, d) in fwd]
    t = np.array([x[1] for x in out]); c = np.array([x[2] for x in out]); sw = np.array([x[3] for x in out]); best = np.array([x[0] for x in out])
    w = t[: len(t) // 64 * 64].reshape(-1, 64)
    res[mode] = best
    print("%-42s tests/ray %6.1f  wave-max %6.1f  candidates/ray %.2f  bound updates/ray %.2f" % (name, t.mean(), w.max(1).mean(), c.mean(), sw.mean()))
print("nearest hits identical with the bound:", bool(np.array_equal(res[0], res[1])), " with the exact culling:", bool(np.array_equal(res[0], res[2])))
