"""Triangle soup: the mesh data of the triangle path as numpy arrays, and the OBJ text that fills it.

The numbers are the ones the reference's loader produces (src/rendering-raycast/model/reader/
obj-reader.ts:23-166 feeding model/triangle.ts:29-45) -- they have to be, the frames depend on them --
but nothing here is shaped like that loader: the text is cut into three attribute tables and one
(triangles, 3 corners) index table in a single pass over the lines, everything after that is whole-array
arithmetic.  What is held per mesh:

    position (T, 3, 3) f64    normal (T, 3, 3) f64    uv (T, 3, 2) f64    color (4,) f64
    centroid (T, 3) f32       -- what the SAH builder sorts by (triangle.ts:37-43 accumulates into a
                                 gl-matrix vec3 = Float32Array: every partial sum is rounded to f32)

Loader behaviour kept, because it decides bits of the upload buffers:
  * a line's kind is its first two characters ('v ', 'vt', 'vn') or its first ('f'); fields are what
    JavaScript's line.split(' ') yields (a doubled blank makes an empty field -> NaN);
  * numbers go through parseFloat / parseInt: the longest numeric prefix, so "0.5\\r" is 0.5;
  * the centre the vertices are shifted by is (min + max) / 2 with the maxima rounded to f32 as they
    are found (vec3.clone -> Float32Array), the minima kept in f64, the sum and the halving each stored
    to f32 (obj-reader.ts:132-164).  Minima / maxima belong to RAW file axes but are subtracted from the
    SWIZZLED position (invertYZ scenes shift raw z by the raw-y centre);
  * alignBottom replaces the offset of raw axis `yIndex` by that axis' minimum; scale multiplies after;
  * a face of k corners is a fan of k - 2 triangles from its first corner whose second and third
    corners are picked through the same y / z swizzle (obj-reader.ts:103-117);
  * uv.v is not flipped here -- the shader does it (RK:387)."""
import re

import numpy as np

_FLOAT_PREFIX = re.compile(r"[ \t\r\n\f\v]*([+-]?(?:Infinity|(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?))")
_INT_PREFIX = re.compile(r"[ \t\r\n\f\v]*([+-]?\d+)")


def js_parse_float(s):
    """JavaScript's parseFloat on a field of split(' ') (None = `undefined`)."""
    if s is None:
        return float("nan")
    m = _FLOAT_PREFIX.match(s)
    if not m:
        return float("nan")
    tok = m.group(1)
    if tok.endswith("Infinity"):
        return float("-inf") if tok[0] == "-" else float("inf")
    return float(tok)


def _js_parse_int(s):
    m = _INT_PREFIX.match(s) if s is not None else None
    return int(m.group(1)) if m else None


def _field(fields, k):
    return fields[k] if k < len(fields) else None


class TriangleSoup:
    def __init__(self, position, normal, uv, color):
        self.position = np.ascontiguousarray(position, dtype=np.float64).reshape(-1, 3, 3)
        self.normal = np.ascontiguousarray(normal, dtype=np.float64).reshape(-1, 3, 3)
        self.uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 3, 2)
        self.color = np.asarray(color, dtype=np.float64).reshape(4)
        self.count = self.position.shape[0]
        # ((f32(0 + p0) + p1 -> f32) + p2 -> f32) / 3 -> f32, each sum formed in f64 (triangle.ts:37-43)
        c = self.position[:, 0, :].astype(np.float32)
        c = (c.astype(np.float64) + self.position[:, 1, :]).astype(np.float32)
        c = (c.astype(np.float64) + self.position[:, 2, :]).astype(np.float32)
        self.centroid = (c.astype(np.float64) / 3.0).astype(np.float32)

    def pack(self):
        """(T, 40) f32: the triangle records of renderer-raytracing.ts:198-209 -- per corner
        {pos.xyz, _, nrm.xyz, _, uv.xy, _, _} at float 12 * corner, the mesh colour at 36."""
        out = np.zeros((self.count, 40), dtype=np.float32)
        for c in range(3):
            out[:, 12 * c:12 * c + 3] = self.position[:, c, :]
            out[:, 12 * c + 4:12 * c + 7] = self.normal[:, c, :]
            out[:, 12 * c + 8:12 * c + 10] = self.uv[:, c, :]
        out[:, 36:40] = self.color
        return out


def parse_obj(text, descriptor=None):
    d = descriptor or {}
    swap_yz = bool(d.get("invertYZ") or False)
    align_bottom = bool(d.get("alignBottom") or False)
    scale = d.get("scale") or 1
    axis = (0, 2, 1) if swap_yz else (0, 1, 2)             # swizzled axis -> raw file axis

    # ---- one pass: attribute tables (raw numbers) and the face table ----
    v_raw, vt_raw, vn_raw = [], [], []
    corner_desc, corner_seen = [], []                      # per fan corner: its "v/vt/vn" text, table sizes at that line
    for line in text.split("\n"):
        head = line[:2]
        if head == "v ":
            f = line.split(" ")
            v_raw.append((js_parse_float(_field(f, 1)), js_parse_float(_field(f, 2)), js_parse_float(_field(f, 3))))
        elif head == "vt":
            f = line.split(" ")
            vt_raw.append((js_parse_float(_field(f, 1)), js_parse_float(_field(f, 2))))
        elif head == "vn":
            f = line.split(" ")
            vn_raw.append((js_parse_float(_field(f, 1)), js_parse_float(_field(f, 2)), js_parse_float(_field(f, 3))))
        elif line[:1] == "f":
            f = line.split(" ")
            seen = (len(v_raw), len(vt_raw), len(vn_raw))
            for i in range(len(f) - 3):
                corner_desc += [_field(f, 1), _field(f, axis[1] + 1 + i), _field(f, axis[2] + 1 + i)]
                corner_seen += [seen, seen, seen]
    v_raw = np.array(v_raw, dtype=np.float64).reshape(-1, 3)
    vt = np.array(vt_raw, dtype=np.float64).reshape(-1, 2)
    vn_raw = np.array(vn_raw, dtype=np.float64).reshape(-1, 3)

    # ---- the shift: per raw axis, f64 minimum and f32 maximum of the cloud ----
    shift = np.zeros(3, dtype=np.float32)
    if len(v_raw):
        lo = v_raw[0].copy()
        hi = v_raw[0].astype(np.float32)
        with np.errstate(invalid="ignore"):
            for a in range(3):
                col = v_raw[:, a]
                col = col[~np.isnan(col)]
                if len(col) and not np.isnan(lo[a]):
                    lo[a] = min(lo[a], col.min())
                    hi[a] = max(hi[a], np.float32(col.max()))
            shift = ((lo + hi.astype(np.float64)).astype(np.float32).astype(np.float64) / 2.0).astype(np.float32)
        if align_bottom:
            shift[axis[1]] = np.float32(lo[axis[1]])
    sw = list(axis)
    v = (v_raw[:, sw] - shift.astype(np.float64)[None, :]) * scale if len(v_raw) else v_raw
    vn = vn_raw[:, sw] if len(vn_raw) else vn_raw

    # ---- faces: "v/vt/vn" -> three index columns, then three gathers ----
    cache = {}
    idx = np.empty((len(corner_desc), 3), dtype=np.int64)
    for k, (desc, seen) in enumerate(zip(corner_desc, corner_seen)):
        got = cache.get(desc)
        if got is None:
            parts = desc.split("/") if desc is not None else []
            got = tuple(_js_parse_int(_field(parts, j)) for j in range(3))
            cache[desc] = got
        for j in range(3):
            if got[j] is None or not (1 <= got[j] <= seen[j]):
                raise ValueError('OBJ face corner "%s" needs v/vt/vn indices that exist' % (desc,))
        idx[k] = got
    idx -= 1
    position = v[idx[:, 0]] if len(idx) else np.zeros((0, 3))
    uv = vt[idx[:, 1]] if len(idx) else np.zeros((0, 2))
    normal = vn[idx[:, 2]] if len(idx) else np.zeros((0, 3))
    return TriangleSoup(position, normal, uv, d.get("color", [0, 0, 0, 0]))
