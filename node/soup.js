'use strict';
// Mesh data of the triangle path as flat typed arrays -- a "soup" -- and the OBJ text that fills it.
//
// What comes out is what the reference's loader produces (src/rendering-raycast/model/reader/
// obj-reader.ts:23-166 feeding model/triangle.ts:29-45), number for number; how it is held is not: no
// Triangle objects, no per-corner arrays -- one Float64Array per attribute, indexed 9 * triangle + 3 *
// corner + axis, and the f32 centroids the SAH builder sorts by in a Float32Array.  Positions, normals
// and texture coordinates stay f64 (the reference keeps them in plain JS arrays until
// renderer-raytracing.ts:198-209 stores them into a Float32Array); centroids are f32 at every step
// (triangle.ts:37-43 adds into a gl-matrix vec3, which is a Float32Array).
//
// Behaviour kept from the reference loader, because the frames depend on it:
//   * a line is classified by its first two characters ('v ', 'vt', 'vn') or its first ('f');
//   * fields are what String.split(' ') yields -- a doubled blank makes an empty field;
//   * the geometric centre that vertices are shifted by is (min + max) / 2 formed in f32, with the
//     maxima themselves kept in f32 (vec3.clone) and the minima in f64 (obj-reader.ts:132-164);
//     minima / maxima are found per SWIZZLED axis but subtracted per RAW axis (invertYZ scenes);
//   * alignBottom replaces the y offset by the (f32) minimum;  scale multiplies after the shift;
//   * a face of k corners becomes k - 2 triangles fanned from its first corner, the second and third
//     corners picked through the same y / z swizzle as coordinates (obj-reader.ts:103-117);
//   * a texture coordinate's v is NOT flipped here (the shader does it, RK:387).

const fround = Math.fround;

// k-th field of `line` in the sense of line.split(' ')[k] (undefined past the end), without building the array
function field(line, k) {
  let start = 0;
  for (let i = 0; i < k; ++i) {
    const sp = line.indexOf(' ', start);
    if (sp < 0) return undefined;
    start = sp + 1;
  }
  const end = line.indexOf(' ', start);
  return end < 0 ? line.slice(start) : line.slice(start, end);
}

function countFields(line) {
  let n = 1;
  for (let p = line.indexOf(' '); p >= 0; p = line.indexOf(' ', p + 1)) ++n;
  return n;
}

class TriangleSoup {
  constructor(position, normal, uv, color) {
    this.count = position.length / 9;
    this.position = Float64Array.from(position);   // [count][3 corners][xyz]
    this.normal = Float64Array.from(normal);       // [count][3 corners][xyz]
    this.uv = Float64Array.from(uv);               // [count][3 corners][uv]
    this.color = Float64Array.from(color);         // rgba of the whole mesh (descriptor.color)
    this.centroid = new Float32Array(3 * this.count);
    for (let t = 0; t < this.count; ++t) {
      for (let a = 0; a < 3; ++a) {
        let c = fround(this.position[9 * t + a]);                 // 0 + corner 0, stored f32
        c = fround(c + this.position[9 * t + 3 + a]);
        c = fround(c + this.position[9 * t + 6 + a]);
        this.centroid[3 * t + a] = c / 3;
      }
    }
  }

  // the 40 floats per triangle of renderer-raytracing.ts:198-209 written into `out` from float index `at`
  packInto(out, at) {
    for (let t = 0; t < this.count; ++t) {
      const loc = at + 40 * t;
      for (let c = 0; c < 3; ++c) {
        for (let a = 0; a < 3; ++a) {
          out[loc + 12 * c + a] = this.position[9 * t + 3 * c + a];
          out[loc + 12 * c + 4 + a] = this.normal[9 * t + 3 * c + a];
        }
        out[loc + 12 * c + 8] = this.uv[6 * t + 2 * c];
        out[loc + 12 * c + 9] = this.uv[6 * t + 2 * c + 1];
      }
      for (let a = 0; a < 4; ++a) out[loc + 36 + a] = this.color[a];
    }
  }
}

function parseObj(text, descriptor) {
  const d = descriptor || {};
  const swapYZ = d.invertYZ ? Boolean(d.invertYZ.valueOf()) : false;
  const alignBottom = d.alignBottom ? Boolean(d.alignBottom.valueOf()) : false;
  const scale = d.scale ? d.scale.valueOf() : 1;
  const axis = swapYZ ? [0, 2, 1] : [0, 1, 2];            // swizzled axis -> raw field
  const lines = text.split('\n');
  const isVertex = (l) => l[0] === 'v' && l[1] === ' ';

  // ---- pass 1: the centre of the vertex cloud ----
  const lo = [0, 0, 0], hi = [0, 0, 0];                  // lo f64, hi f32 (see header)
  for (const l of lines) {
    if (!isVertex(l)) continue;
    for (let a = 0; a < 3; ++a) { lo[a] = parseFloat(field(l, 1 + a)); hi[a] = fround(lo[a]); }
    break;
  }
  for (const l of lines) {
    if (!isVertex(l)) continue;
    for (let s = 0; s < 3; ++s) {
      const v = parseFloat(field(l, 1 + axis[s]));
      if (v < lo[axis[s]]) lo[axis[s]] = v;
      if (v > hi[axis[s]]) hi[axis[s]] = fround(v);
    }
  }
  const shift = [0, 0, 0];
  for (let a = 0; a < 3; ++a) shift[a] = fround(fround(lo[a] + hi[a]) / 2);
  if (alignBottom) shift[axis[1]] = fround(lo[axis[1]]);

  // ---- pass 2: attributes, then faces into the soup ----
  const vx = [], vt = [], vn = [];                        // flat: 3 / 2 / 3 numbers per entry
  const position = [], normal = [], uv = [];
  const corner = (desc) => {
    const s1 = desc.indexOf('/'), s2 = s1 < 0 ? -1 : desc.indexOf('/', s1 + 1);
    const iv = parseInt(s1 < 0 ? desc : desc.slice(0, s1), 10) - 1;
    const it = s1 < 0 ? NaN : parseInt(s2 < 0 ? desc.slice(s1 + 1) : desc.slice(s1 + 1, s2), 10) - 1;
    const inn = s2 < 0 ? NaN : parseInt(desc.slice(s2 + 1), 10) - 1;
    if (!(iv >= 0 && 3 * iv < vx.length && it >= 0 && 2 * it < vt.length && inn >= 0 && 3 * inn < vn.length))
      throw new Error('OBJ face corner "' + desc + '" needs v/vt/vn indices that exist');
    position.push(vx[3 * iv], vx[3 * iv + 1], vx[3 * iv + 2]);
    uv.push(vt[2 * it], vt[2 * it + 1]);
    normal.push(vn[3 * inn], vn[3 * inn + 1], vn[3 * inn + 2]);
  };
  for (const l of lines) {
    if (isVertex(l)) {
      for (let s = 0; s < 3; ++s) vx.push((parseFloat(field(l, 1 + axis[s])) - shift[s]) * scale);
    } else if (l[0] === 'v' && l[1] === 't') {
      vt.push(parseFloat(field(l, 1)), parseFloat(field(l, 2)));
    } else if (l[0] === 'v' && l[1] === 'n') {
      for (let s = 0; s < 3; ++s) vn.push(parseFloat(field(l, 1 + axis[s])));
    } else if (l[0] === 'f') {
      const fan = countFields(l) - 3;
      for (let i = 0; i < fan; ++i) {
        corner(field(l, 1));
        corner(field(l, axis[1] + 1 + i));
        corner(field(l, axis[2] + 1 + i));
      }
    }
  }
  return new TriangleSoup(position, normal, uv, d.color || [0, 0, 0, 0]);
}

module.exports = { TriangleSoup, parseObj };
