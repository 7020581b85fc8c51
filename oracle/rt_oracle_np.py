"""rt_oracle_np.py -- an INDEPENDENT numpy restatement of the reference shader (sphere path),
vectorised over pixels, written separately from oracle/rt_oracle.c to catch restatement bugs.

TEST INFRASTRUCTURE ONLY; parity: see rt_oracle.h (the C oracle is pinned by the reference's screenshot; this twin follows it bit for bit).
It cannot pin the C oracle to the reference; it only shows that two separate readings of
the WGSL agree bit for bit.  All arithmetic is numpy float32 (IEEE single operations, no
fusion), in the order the WGSL grammar gives.

Citations relative to /root/reference/:
  RK = src/rendering-raycast/shaders/raytracer-kernel.wgsl
  HK = src/rendering-raycast/shaders/heatmap-kernel.wgsl
"""
import numpy as np

F = np.float32


def _dot(ax, ay, az, bx, by, bz):
    return (ax * bx + ay * by) + az * bz


def _normalize(x, y, z):
    ln = np.sqrt(_dot(x, y, z, x, y, z))
    return x / ln, y / ln, z / ln


def _face_point(f, S, T, n):
    """3-D point (integer, face planes at +-n) of the texel centre S = 2i+1-n, T = 2j+1-n of face f:
    Vulkan's cube face table (+X: sc=-z tc=-y; -X: sc=+z tc=-y; +Y: sc=+x tc=+z; -Y: sc=+x tc=-z;
    +Z: sc=+x tc=-y; -Z: sc=-x tc=-y) solved for (x, y, z)."""
    N = np.full_like(S, n)
    return {0: (N, -T, -S), 1: (-N, -T, S), 2: (S, N, T), 3: (S, -N, -T), 4: (S, -T, N), 5: (-S, -T, -N)}[f]


_NEIGHBOUR_CACHE = {}


def _edge_neighbours(n):
    """For every face and each of its four edges: the (face', i', j') of the texel across the edge,
    per position along it.  Found WITHOUT the fold arithmetic of rt_oracle.c: the centre of a tap one
    step outside face f lies on f's extended plane; the texel it stands for is the one of another
    face whose centre is nearest to that point (distance sqrt(2) texel half-widths, unique)."""
    if n in _NEIGHBOUR_CACHE:
        return _NEIGHBOUR_CACHE[n]
    idx = np.arange(n)
    border = []           # (face, i, j, point) of every texel in the outermost ring of every face
    for f in range(6):
        ii = np.concatenate([idx, idx, np.zeros(n, int), np.full(n, n - 1)])
        jj = np.concatenate([np.zeros(n, int), np.full(n, n - 1), idx, idx])
        x, y, z = _face_point(f, 2 * ii + 1 - n, 2 * jj + 1 - n, n)
        border.append((np.full(ii.shape, f), ii, jj, np.stack([x, y, z], 1).astype(np.int64)))
    table = {}
    for f in range(6):
        others = [b for k, b in enumerate(border) if k != f]
        of = np.concatenate([b[0] for b in others]); oi = np.concatenate([b[1] for b in others])
        oj = np.concatenate([b[2] for b in others]); op = np.concatenate([b[3] for b in others])
        for side, (ti, tj) in {"left": (np.full(n, -1), idx), "right": (np.full(n, n), idx),
                                "top": (idx, np.full(n, -1)), "bottom": (idx, np.full(n, n))}.items():
            x, y, z = _face_point(f, 2 * ti + 1 - n, 2 * tj + 1 - n, n)
            q = np.stack([x, y, z], 1).astype(np.int64)
            d2 = ((q[:, None, :] - op[None, :, :]) ** 2).sum(-1)
            k = d2.argmin(1)
            assert (d2[np.arange(n), k] == 2).all()
            table[(f, side)] = (of[k], oi[k], oj[k])
    _NEIGHBOUR_CACHE[n] = table
    return table


def _cube(faces, rx, ry, rz):
    """textureSampleLevel(skyTex, texSamp, dir, 0).rgb (RK:92,123): major-axis face selection
    (z wins ties over y over x), bilinear, lerp a + (b-a)*f.  Six equal square faces (a WebGPU cube
    texture) filter seamlessly across edges, corners take a + ((b-a)+(c-a))/3 of the three texels
    that meet there (Vulkan "Cube Map Edge / Corner Handling"); anything else clamps to the edge of
    the selected image."""
    ax, ay, az = np.abs(rx), np.abs(ry), np.abs(rz)
    isz = (az >= ax) & (az >= ay)
    isy = (~isz) & (ay >= ax)
    isx = ~(isz | isy)
    face = np.zeros(rx.shape, np.int32)
    sc = np.zeros_like(rx); tc = np.zeros_like(rx); ma = np.ones_like(rx)
    pz = isz & (rz >= 0); nz = isz & ~(rz >= 0)
    py = isy & (ry >= 0); ny = isy & ~(ry >= 0)
    px = isx & (rx >= 0); nx = isx & ~(rx >= 0)
    for m, f, s_, t_, a_ in ((pz, 4, rx, -ry, az), (nz, 5, -rx, -ry, az), (py, 2, rx, rz, ay),
                             (ny, 3, rx, -rz, ay), (px, 0, -rz, -ry, ax), (nx, 1, rz, -ry, ax)):
        face[m] = f; sc[m] = s_[m]; tc[m] = t_[m]; ma[m] = a_[m]
    out = np.zeros(rx.shape + (3,), F)
    s = F(0.5) * (sc / ma) + F(0.5)
    t = F(0.5) * (tc / ma) + F(0.5)
    imgs = [np.asarray(f_, np.uint8) for f_ in faces]
    n0 = imgs[0].shape[1]
    seamless = all(im.shape[0] == n0 and im.shape[1] == n0 for im in imgs)
    stack = np.stack([im[..., :3] for im in imgs]) if seamless else None      # [6][n][n][3]
    nb = _edge_neighbours(n0) if seamless else None
    for f in range(6):
        m = face == f
        if not m.any():
            continue
        img = imgs[f]
        h, w = img.shape[:2]
        u = s[m] * F(w) - F(0.5)
        v = t[m] * F(h) - F(0.5)
        fu, fv = np.floor(u), np.floor(v)
        wu, wv = (u - fu)[:, None], (v - fv)[:, None]
        x0, y0 = fu.astype(np.int64), fv.astype(np.int64)

        def tex(xx, yy):
            if not seamless:
                xx = np.clip(xx, 0, w - 1); yy = np.clip(yy, 0, h - 1)
                return img[yy, xx, :3].astype(F) / F(255.0)
            cx, cy = np.clip(xx, 0, w - 1), np.clip(yy, 0, w - 1)
            own = stack[f, cy, cx].astype(F) / F(255.0)

            def across(side_lo, side_hi, coord, along):
                ff, ii, jj = np.zeros_like(xx), np.zeros_like(xx), np.zeros_like(xx)
                for side, sel in ((side_lo, coord < 0), (side_hi, coord >= w)):
                    tf, ti, tj = nb[(f, side)]
                    ff[sel] = tf[along[sel]]; ii[sel] = ti[along[sel]]; jj[sel] = tj[along[sel]]
                return stack[ff, jj, ii].astype(F) / F(255.0)
            ox, oy = (xx < 0) | (xx >= w), (yy < 0) | (yy >= w)
            res = own.copy()
            eu = across("left", "right", xx, cy)      # texel across the u edge, at the (clamped) row
            ev = across("top", "bottom", yy, cx)      # texel across the v edge, at the (clamped) column
            only_u, only_v, both = ox & ~oy, oy & ~ox, ox & oy
            res[only_u] = eu[only_u]
            res[only_v] = ev[only_v]
            res[both] = (own + ((eu - own) + (ev - own)) / F(3.0))[both]
            return res
        c00, c10, c01, c11 = tex(x0, y0), tex(x0 + 1, y0), tex(x0, y0 + 1), tex(x0 + 1, y0 + 1)
        top = c00 + wu * (c10 - c00)
        bot = c01 + wu * (c11 - c01)
        out[m] = top + wv * (bot - top)
    return out


def _trace(spheres, ox, oy, oz, dx, dy, dz):
    """brute-force nearest hit (HK:307-331 called as RK:311-322: tMin 0.001, tMax running nearest)."""
    n = ox.shape[0]
    nearest = np.full(n, F(9999.0), F)
    idx = np.full(n, -1, np.int64)
    a = _dot(dx, dy, dz, dx, dy, dz)
    for i in range(spheres.shape[0]):
        cx, cy, cz, radius = spheres[i, 0], spheres[i, 1], spheres[i, 2], spheres[i, 7]
        ocx, ocy, ocz = ox - cx, oy - cy, oz - cz
        b = F(2.0) * _dot(dx, dy, dz, ocx, ocy, ocz)
        c = _dot(ocx, ocy, ocz, ocx, ocy, ocz) - radius * radius
        disc = b * b - F(4.0) * a * c
        with np.errstate(invalid="ignore"):
            t = (-b - np.sqrt(disc)) / (F(2.0) * a)
            hit = (disc > 0) & (t > F(0.001)) & (t < nearest)
        nearest = np.where(hit, t, nearest)
        idx = np.where(hit, i, idx)
    return nearest, idx


def render(params, spheres, faces, W, H, want_rays=False):
    """Returns (rgba8 (H,W,4) uint8, rgb float32 (H,W,3), total rays)."""
    p = np.asarray(params, F)
    sp = np.asarray(spheres, F).reshape(-1, 8)
    cam, fw, rt, up, L = p[0:3], p[4:7], p[8:11], p[12:15], p[16:19]
    Li, minI, mb = p[19], p[20], p[21]
    bounces = int(mb) if mb > 0 else 0
    ys, xs = np.mgrid[0:H, 0:W]
    xs = xs.reshape(-1).astype(np.int32); ys = ys.reshape(-1).astype(np.int32)
    hc = (xs.astype(F) - F(W) / F(2)) / F(W) * F(2)                      # RK:78
    vc = (F(H) / F(2) - ys.astype(F)) / F(W) * F(2)                      # RK:79
    d0 = [(fw[k] + hc * rt[k]) + vc * up[k] for k in range(3)]
    d0x, d0y, d0z = _normalize(*d0)                                       # RK:82-86
    n = xs.shape[0]
    ox = np.full(n, cam[0], F); oy = np.full(n, cam[1], F); oz = np.full(n, cam[2], F)
    dx, dy, dz = d0x.copy(), d0y.copy(), d0z.copy()
    color = np.ones((n, 3), F)
    dist = np.zeros(n, F)
    affect = F(1.0); ssum = F(0.0)
    alive = np.ones(n, bool)
    rays = 0
    for bounce in range(bounces):                                         # RK:113
        ia = np.nonzero(alive)[0]
        if ia.size == 0:
            break
        t, idx = _trace(sp, ox[ia], oy[ia], oz[ia], dx[ia], dy[ia], dz[ia])   # RK:114
        rays += ia.size
        hit = idx >= 0
        if bounce == 0:
            dist[ia] = np.where(hit, t, F(0.0))                           # RK:116-118
        nxt = F(affect + ssum)                                            # RK:120
        miss = ia[~hit]
        if miss.size:                                                     # RK:122-126
            sky = _cube(faces, dx[miss], dy[miss], dz[miss]) * minI
            color[miss] = (color[miss] * ssum + sky * affect) / nxt
            alive[miss] = False
        hi = ia[hit]
        if hi.size:
            th, ih = t[hit], idx[hit]
            px_, py_, pz_ = ox[hi] + th * dx[hi], oy[hi] + th * dy[hi], oz[hi] + th * dz[hi]   # RK:129
            nx, ny, nz = _normalize(px_ - sp[ih, 0], py_ - sp[ih, 1], pz_ - sp[ih, 2])         # HK:320
            k2 = F(2.0) * _dot(nx, ny, nz, dx[hi], dy[hi], dz[hi])                             # reflect
            rx, ry, rz = _normalize(dx[hi] - k2 * nx, dy[hi] - k2 * ny, dz[hi] - k2 * nz)      # RK:130
            ox[hi], oy[hi], oz[hi] = px_, py_, pz_
            dx[hi], dy[hi], dz[hi] = rx, ry, rz
            # lightIntensity, RK:146-166
            sx, sy, sz = _normalize(px_ - L[0], py_ - L[1], pz_ - L[2])
            distance = np.sqrt(_dot(sx, sy, sz, sx, sy, sz))
            st, sidx = _trace(sp, np.full(hi.size, L[0], F), np.full(hi.size, L[1], F), np.full(hi.size, L[2], F),
                              sx, sy, sz)
            rays += hi.size
            hx_, hy_, hz_ = L[0] + st * sx, L[1] + st * sy, L[2] + st * sz
            ex, ey, ez = hx_ - px_, hy_ - py_, hz_ - pz_
            diff = np.sqrt(_dot(ex, ey, ez, ex, ey, ez))
            lit = (sidx >= 0) & (diff < F(0.005))
            power = np.minimum(np.maximum(_dot(nx, ny, nz, -sx, -sy, -sz), minI), F(1.0))
            cap = Li / (Li + distance)
            inten = np.where(lit, power * cap, minI).astype(F)
            blended = sp[ih, 4:7] * inten[:, None]
            color[hi] = (color[hi] * ssum + blended * affect) / nxt       # RK:136
        affect = F(affect / F(2.0))                                       # RK:139
        ssum = nxt                                                        # RK:140
    sky0 = _cube(faces, d0x, d0y, d0z) * minI                             # RK:92
    k = np.minimum(np.maximum((F(30.0) - dist) / F(30.0), F(0.0)), F(1.0))[:, None]   # RK:95
    pix = color * k + sky0 * (F(1.0) - k)                                 # RK:96
    q = np.floor(np.minimum(np.maximum(pix, F(0.0)), F(1.0)) * F(255.0) + F(0.5)).astype(np.uint8)
    rgba = np.concatenate([q, np.full((n, 1), 255, np.uint8)], axis=1).reshape(H, W, 4)
    return rgba, pix.reshape(H, W, 3).astype(F), rays
