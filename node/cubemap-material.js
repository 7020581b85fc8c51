'use strict';
// CubemapMaterial -- src/material/cubemap-material.ts:1-80: six rgba8unorm faces in the order
// +X,-X,+Y,-Y,+Z,-Z.  Image decoding is the caller's business (the browser does it in the
// reference); faces are {width, height, data: Uint8Array}.
class CubemapMaterial {
  constructor() { this.faces = []; }
  static constant(rgba) {
    const m = new CubemapMaterial();
    for (let i = 0; i < 6; ++i) m.faces.push({ width: 1, height: 1, data: new Uint8Array(rgba) });
    return m;
  }
  static fromCross(width, height, data) {                            // cubemap-material.ts:35-58
    if (width % 4 || height % 3) throw new Error('fromCross: width must divide by 4 and height by 3');
    const sw = width / 4, sh = height / 3;
    const positions = [[2, 1], [0, 1], [1, 0], [1, 2], [1, 1], [3, 1]];
    const m = new CubemapMaterial();
    for (const [c, r] of positions) {
      const face = new Uint8Array(sw * sh * 4);
      for (let y = 0; y < sh; ++y) {
        const src = ((r * sh + y) * width + c * sw) * 4;
        face.set(data.subarray(src, src + sw * 4), y * sw * 4);
      }
      m.faces.push({ width: sw, height: sh, data: face });
    }
    return m;
  }
}
module.exports = { CubemapMaterial };
