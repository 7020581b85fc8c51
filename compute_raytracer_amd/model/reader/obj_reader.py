"""ObjectReader -- src/rendering-raycast/model/reader/obj-reader.ts:5-167, quirks included:

* the reader's fields are STATIC, shared across loads (obj-reader.ts:6-20): `mins`/`maxs`/
  `offsets` of the previous file survive until initMinMax overwrites them;
* initMinMax seeds mins/maxs from the first vertex line's components in FILE order
  (obj-reader.ts:136-141) but updates them through the swizzled indices (obj-reader.ts:149-158);
* faces are fan-triangulated around the first vertex with the yIndex/zIndex swizzle applied to
  the FACE corner positions as well (obj-reader.ts:103-117), so invertYZ also flips the winding;
* components are split on single spaces (`line.split(' ')`), so doubled spaces shift fields;
* all three of v/vt/vn indices are required (`parseInt(undefined)` is NaN -> undefined corner).

`fetch(url)` becomes reading a file or a string; parseFloat becomes float().
"""
import math

from ... import glmatrix as glm
from ..triangle import Triangle


def _parse_float(s):
    """JS parseFloat: leading numeric prefix, NaN if none."""
    s = s.strip()
    end = 0
    seen_digit = seen_dot = seen_exp = False
    i = 0
    if i < len(s) and s[i] in "+-":
        i += 1
    while i < len(s):
        ch = s[i]
        if ch.isdigit():
            seen_digit = True
            end = i + 1
        elif ch == "." and not seen_dot and not seen_exp:
            seen_dot = True
        elif ch in "eE" and seen_digit and not seen_exp:
            seen_exp = True
            if i + 1 < len(s) and s[i + 1] in "+-":
                i += 1
        else:
            break
        i += 1
    try:
        return float(s[:end]) if seen_digit else math.nan
    except ValueError:
        return math.nan


def _parse_int(s):
    try:
        return int(_parse_float(s))
    except (ValueError, OverflowError):
        return None


class ObjectReader:
    # static state, obj-reader.ts:6-20
    color = None
    v, vt, vn = [], [], []
    mins, maxs, offsets = [0, 0, 0], [0, 0, 0], [0, 0, 0]
    alignBottom = False
    scale = 1
    xIndex, yIndex, zIndex = 0, 1, 2

    @classmethod
    def loadMeshFromObjText(cls, text, descriptor):            # obj-reader.ts:23-44
        cls.color = descriptor["color"]
        invertYZ = bool(descriptor.get("invertYZ", False))
        cls.alignBottom = bool(descriptor.get("alignBottom", False))
        cls.scale = descriptor.get("scale") or 1
        if invertYZ:
            cls.yIndex, cls.zIndex = 2, 1
        else:
            cls.yIndex, cls.zIndex = 1, 2
        return cls.createMeshFromText(text)

    @classmethod
    def loadMeshFromObjFile(cls, path, descriptor):
        with open(path, "r") as f:
            return cls.loadMeshFromObjText(f.read(), descriptor)

    @classmethod
    def createMeshFromText(cls, fileContent):                  # obj-reader.ts:46-69
        triangles = []
        lines = fileContent.split("\n")
        cls.initMinMax(lines)
        for line in lines:
            c0 = line[0] if len(line) > 0 else None
            c1 = line[1] if len(line) > 1 else None
            if c0 == "v" and c1 == " ": cls.readVertexLine(line)
            elif c0 == "v" and c1 == "t": cls.readTexcoordLine(line)
            elif c0 == "v" and c1 == "n": cls.readNormalLine(line)
            elif c0 == "f": cls.addTriangleFromFaceData(line, triangles)
        cls.v, cls.vt, cls.vn = [], [], []
        return triangles

    @classmethod
    def readVertexLine(cls, line):                             # obj-reader.ts:71-82
        c = line.split(" ")
        v = [_parse_float(c[1 + cls.xIndex]), _parse_float(c[1 + cls.yIndex]), _parse_float(c[1 + cls.zIndex])]
        glm.vec3_subtract(v, v, cls.offsets)
        glm.vec3_mul(v, v, [cls.scale, cls.scale, cls.scale])
        cls.v.append(v)

    @classmethod
    def readTexcoordLine(cls, line):                           # obj-reader.ts:84-91
        c = line.split(" ")
        cls.vt.append([_parse_float(c[1]), _parse_float(c[2])])

    @classmethod
    def readNormalLine(cls, line):                             # obj-reader.ts:93-101
        c = line.split(" ")
        cls.vn.append([_parse_float(c[1 + cls.xIndex]), _parse_float(c[1 + cls.yIndex]), _parse_float(c[1 + cls.zIndex])])

    @classmethod
    def addTriangleFromFaceData(cls, line, triangles):         # obj-reader.ts:103-117
        line = line.replace("\n", "", 1)
        desc = line.split(" ")
        triangleCount = len(desc) - 3
        for i in range(triangleCount):
            t = Triangle()
            t.color = cls.color
            cls.readCorner(desc[1], t)
            cls.readCorner(desc[cls.yIndex + 1 + i], t)
            cls.readCorner(desc[cls.zIndex + 1 + i], t)
            t.calculateCentroid()
            triangles.append(t)

    @classmethod
    def readCorner(cls, vertexDescription, triangle):          # obj-reader.ts:119-130
        parts = vertexDescription.split("/")

        def pick(arr, k):
            if k >= len(parts):
                return None
            n = _parse_int(parts[k])
            if n is None or n - 1 < 0 or n - 1 >= len(arr):
                return None
            return arr[n - 1]
        triangle.corners.append(pick(cls.v, 0))
        triangle.normals.append(pick(cls.vn, 2))
        triangle.textures.append(pick(cls.vt, 1))

    @classmethod
    def initMinMax(cls, lines):                                # obj-reader.ts:132-166
        for line in lines:
            if len(line) > 1 and line[0] == "v" and line[1] == " ":
                c = line.split(" ")
                cls.mins = [_parse_float(c[1]), _parse_float(c[2]), _parse_float(c[3])]
                cls.maxs = glm.vec3_from_values(*cls.mins)     # vec3.clone -> Float32Array
                break
        for line in lines:
            if len(line) > 1 and line[0] == "v" and line[1] == " ":
                c = line.split(" ")
                x = _parse_float(c[1 + cls.xIndex]); y = _parse_float(c[1 + cls.yIndex]); z = _parse_float(c[1 + cls.zIndex])
                if x < cls.mins[cls.xIndex]: cls.mins[cls.xIndex] = x
                if y < cls.mins[cls.yIndex]: cls.mins[cls.yIndex] = y
                if z < cls.mins[cls.zIndex]: cls.mins[cls.zIndex] = z
                if x > float(cls.maxs[cls.xIndex]): cls.maxs[cls.xIndex] = x
                if y > float(cls.maxs[cls.yIndex]): cls.maxs[cls.yIndex] = y
                if z > float(cls.maxs[cls.zIndex]): cls.maxs[cls.zIndex] = z
        cls.offsets = glm.vec3_add(glm.vec3_create(), cls.mins, cls.maxs)
        glm.vec3_div(cls.offsets, cls.offsets, [2, 2, 2])
        if cls.alignBottom:
            cls.offsets[cls.yIndex] = cls.mins[cls.yIndex]
