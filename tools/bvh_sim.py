"""Design study (CPU, float64): tests per ray of the threaded bounding-sphere hierarchy of
rt_bvh.hip for different build strategies, on rays of a BASELINE scene (primary, reflection and
shadow rays of a sample of 8x8 tiles).  Reports mean tests per ray and the per-wave maximum
(what a wave of 64 lanes actually pays).  Not product code.
usage: python tools/bvh_sim.py [C3] [ntiles]"""
import sys, os, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compute_raytracer_amd.scene_raytracing import synthetic_spheres, BASELINE_CONFIGS, SceneRaytracing

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
ntiles = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = BASELINE_CONFIGS[cfgname]
sph = synthetic_spheres(cfg["spheres"], cfg["seed"])
C = np.array([s.center for s in sph], dtype=np.float64)
R = np.array([s.radius for s in sph], dtype=np.float64)
N = len(R)
SIGMA = float(os.environ.get('SIGMA', '1.16'))
KAPPA = float(os.environ.get('KAPPA', str(2.0 ** -16)))


def bound_box(ids):
    lo = (C[ids] - R[ids, None]).min(0); hi = (C[ids] + R[ids, None]).max(0)
    bc = 0.5 * (lo + hi)
    return bc, (np.linalg.norm(C[ids] - bc, axis=1) + R[ids]).max()


def bound_ritter(ids):
    bc, br = bound_box(ids)
    # shrink-wrap iterations: move the centre towards the farthest member
    for _ in range(32):
        d = np.linalg.norm(C[ids] - bc, axis=1) + R[ids]
        k = int(np.argmax(d))
        step = (C[ids[k]] - bc)
        nrm = np.linalg.norm(step)
        if nrm < 1e-12: break
        cand = bc + step / nrm * 0.05 * d[k]
        r2 = (np.linalg.norm(C[ids] - cand, axis=1) + R[ids]).max()
        if r2 < br: bc, br = cand, r2
        else: break
    return bc, br


def split_median(ids, bound):
    lo = C[ids].min(0); hi = C[ids].max(0)
    ax = int(np.argmax(hi - lo))
    o = ids[np.argsort(C[ids, ax], kind="stable")]
    h = len(o) // 2
    return o[:h], o[h:]


def split_sah(ids, bound):
    best = None
    for ax in range(3):
        o = ids[np.argsort(C[ids, ax], kind="stable")]
        n = len(o)
        for h in range(1, n):
            if n > 16 and h % max(1, n // 16): continue
            _, ra = bound(o[:h]); _, rb = bound(o[h:])
            cost = ra * ra * h + rb * rb * (n - h)
            if best is None or cost < best[0]: best = (cost, o[:h], o[h:])
    return best[1], best[2]


def split_sah_diag(ids, bound):
    # cost proxy: squared half-diagonal of the members' box (radii included) times the count,
    # every split position of every axis, prefix/suffix boxes
    best = None
    for ax in range(3):
        o = ids[np.argsort(C[ids, ax], kind="stable")]
        lo = C[o] - R[o, None]; hi = C[o] + R[o, None]
        plo = np.minimum.accumulate(lo, 0); phi = np.maximum.accumulate(hi, 0)
        slo = np.minimum.accumulate(lo[::-1], 0)[::-1]; shi = np.maximum.accumulate(hi[::-1], 0)[::-1]
        n = len(o)
        h = np.arange(1, n)
        ca = ((phi[:-1] - plo[:-1]) ** 2).sum(1) * h
        cb = ((shi[1:] - slo[1:]) ** 2).sum(1) * (n - h)
        k = int(np.argmin(ca + cb))
        if best is None or ca[k] + cb[k] < best[0]: best = (ca[k] + cb[k], o[:k + 1], o[k + 1:])
    return best[1], best[2]


def build_collapsed(split, bound, arity=4):
    """binary tree by `split` down to single spheres, collapsed to `arity` by opening the child with the
    largest radius first; same threaded layout as build()"""
    def bin_tree(ids):
        if len(ids) == 1: return ("l", ids[0])
        a, b = split(ids, bound)
        return ("n", bin_tree(a), bin_tree(b), ids)
    rec = []; link = []
    def leaf(i):
        rec.append((C[i], R[i])); link.append(("l", i))
    def kids(node):
        ch = [node[1], node[2]]
        while len(ch) < arity:
            best = -1; br = -1.0
            for k, c in enumerate(ch):
                if c[0] == "n":
                    r = bound(c[3])[1]
                    if r > br: br = r; best = k
            if best < 0: break
            c = ch.pop(best); ch[best:best] = [c[1], c[2]]
        return ch
    def emit(node):
        if node[0] == "l": leaf(node[1]); return
        me = len(rec); rec.append(None); link.append(None)
        for c in kids(node): emit(c)
        bc, br = bound(node[3])
        rec[me] = (bc, br * SIGMA); link[me] = ("n", len(rec))
    med = np.median(R); ext = np.linalg.norm(C.max(0) - C.min(0))
    rest = []
    for i in range(N):
        if N > 8 and R[i] > 8 * med and R[i] > 0.125 * ext: leaf(i)
        else: rest.append(i)
    root = bin_tree(np.array(rest))
    for c in kids(root): emit(c)
    return rec, link


GROUP = False      # leaves of up to `leafmax` spheres under ONE bound, members appended untested (round 4 study)
def build(split, bound, leafmax=4, arity=4):
    rec = []; link = []
    def leaf(i):
        rec.append((C[i], R[i])); link.append(("l", i))
    def emit(ids):
        if len(ids) == 1: leaf(ids[0]); return
        if GROUP and len(ids) <= leafmax:
            bc, br = bound(ids)
            rec.append((bc, br * SIGMA)); link.append(("g", ids)); return
        me = len(rec); rec.append(None); link.append(None)
        children(ids)
        bc, br = bound(ids)
        rec[me] = (bc, br * SIGMA); link[me] = ("n", len(rec))
    def children(ids):
        if len(ids) <= leafmax and not GROUP:
            for i in ids: leaf(i)
            return
        parts = [ids]
        while len(parts) < arity:
            parts.sort(key=len, reverse=True)
            big = parts.pop(0)
            if len(big) < 2: parts.append(big); break
            a, b = split(big, bound)
            parts += [a, b]
        for p in parts: emit(p)
    med = np.median(R)
    ext = np.linalg.norm(C.max(0) - C.min(0))
    rest = []
    for i in range(N):
        if N > 8 and R[i] > 8 * med and R[i] > 0.125 * ext: leaf(i)
        else: rest.append(i)
    children(np.array(rest))
    return rec, link


TCULL = False
def traverse(rec, link, o, d):
    i = 0; n = len(rec); tests = 0; cands = 0
    best = 1e30; bi = -1
    while i < n:
        c, r = rec[i]; kind, x = link[i]
        oc = o - c
        b = oc @ d; cc = oc @ oc - r * r
        bm = min(b, 0.0)
        ok = bm * bm * (1 + KAPPA) ** 2 - cc + (r * r * KAPPA if kind == 'l' else 0.0) > 0
        if TCULL and ok and cc > 0 and best < 1e29:
            ok = (-b - math.sqrt(b * b - cc)) < best        # entry distance beyond the nearest hit so far
        tests += 1
        if kind == "g":
            if ok:
                for s_ in x:
                    cands += 1
                    oc2 = o - C[s_]; b2 = oc2 @ d; c2 = oc2 @ oc2 - R[s_] * R[s_]
                    disc = b2 * b2 - c2
                    if disc > 0 and b2 < 0:
                        t = -b2 - math.sqrt(disc)
                        if t > 1e-3 and t < best: best, bi = t, s_
            i += 1
        elif kind == "l":
            if ok:
                cands += 1
                disc = b * b - cc
                if disc > 0 and b < 0:
                    t = -b - math.sqrt(disc)
                    if t > 1e-3 and t < best: best, bi = t, x
            i += 1
        else:
            i = i + 1 if ok else x
    return best, bi, tests, cands


sc = SceneRaytracing().createScene(sph)
cam = sc.camera
W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
light = np.array(sc.light.position, dtype=np.float64)
fw = np.array(cam.forwards, dtype=np.float64); rt = np.array(cam.right, dtype=np.float64); up = np.array(cam.up, dtype=np.float64)
cp = np.array(cam.position, dtype=np.float64)
rng = np.random.default_rng(1)
tx = rng.integers(0, W // 8, ntiles); ty = rng.integers(0, H // 8, ntiles)

# rays once, with a reference hierarchy
def gen_rays(rec, link):
    rays = []
    for t in range(ntiles):
        lanes = []
        for ly in range(8):
            for lx in range(8):
                x = tx[t] * 8 + lx; y = ty[t] * 8 + ly
                hc = (x - W / 2) / W * 2; vc = (H / 2 - y) / W * 2
                d = fw + hc * rt + vc * up; d /= np.linalg.norm(d)
                lanes.append((cp.copy(), d))
        for b in range(B):
            nxt = []
            for (o, d) in lanes:
                rays.append((t, o, d))
                tt, i, _, _ = traverse(rec, link, o, d)
                if i < 0: continue
                p = o + tt * d; n = (p - C[i]) / R[i]
                sd = p - light; sd /= np.linalg.norm(sd)
                rays.append((t, light, sd))
                d2 = d - 2 * (d @ n) * n
                nxt.append((p, d2 / np.linalg.norm(d2)))
            lanes = nxt
    return rays

rec0, link0 = build(split_median, bound_box)
rays = gen_rays(rec0, link0)
print(cfgname, "N", N, "rays", len(rays))
import itertools
for (name, split, bound, leafmax, arity), tc in itertools.product([
        ("median/box   4/4", split_median, bound_box, 4, 4),
        ("median/ritter4/4", split_median, bound_ritter, 4, 4),
        ("sah/box      4/4", split_sah, bound_box, 4, 4),
        ("sah/ritter   4/4", split_sah, bound_ritter, 4, 4),
        ("sahdiag/box  4/4", split_sah_diag, bound_box, 4, 4),
        ("sahdiag/ritt 4/4", split_sah_diag, bound_ritter, 4, 4),
        ("sahdiag/box  3/3", split_sah_diag, bound_box, 3, 3),
        ("sahdiag/box  6/4", split_sah_diag, bound_box, 6, 4),
        ("collapse4 sahdiag", split_sah_diag, bound_ritter, 0, 4),
        ("collapse3 sahdiag", split_sah_diag, bound_ritter, 0, 3),
        ("collapse6 sahdiag", split_sah_diag, bound_ritter, 0, 6),
        ("collapse2 sahdiag", split_sah_diag, bound_ritter, 0, 2),
        ("GROUP sahdiag/ritt 2/4", split_sah_diag, bound_ritter, 2, 4),
        ("GROUP sahdiag/ritt 3/4", split_sah_diag, bound_ritter, 3, 4),
        ("GROUP sahdiag/ritt 4/4", split_sah_diag, bound_ritter, 4, 4)][5:], [False]):
    if os.environ.get('ONLY') and os.environ['ONLY'] not in name: continue
    TCULL = tc
    GROUP = name.startswith('GROUP')
    rec, link = build(split, bound, leafmax, arity) if leafmax else build_collapsed(split, bound, arity)
    tests = np.array([traverse(rec, link, o, d)[2] for (_, o, d) in rays])
    cands = np.array([traverse(rec, link, o, d)[3] for (_, o, d) in rays[::7]])
    # waves: consecutive groups of 64 rays (what regeneration approximates)
    w = tests[: len(tests) // 64 * 64].reshape(-1, 64)
    print("tcull" if tc else "     ", "%-18s nodes %5d  tests/ray %6.1f  wave-max %6.1f  lane-eff %4.1f%%  cands/ray %.2f" %
          (name, len(rec), tests.mean(), w.max(1).mean(), 100 * w.mean() / w.max(1).mean(), cands.mean()))
