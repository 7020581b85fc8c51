'use strict';
// ObjectReader -- src/rendering-raycast/model/reader/obj-reader.ts:5-167, static state and quirks
// included (see the Python mirror's header for the list).  `fetch(url)` becomes a string or a file.
const fs = require('fs');
const { vec3 } = require('../../gl-matrix-lite');
const { Triangle } = require('../triangle');

class ObjectReader {
  static loadMeshFromObjText(text, descriptor) {                       // obj-reader.ts:23-44
    this.color = descriptor.color;
    const invertYZ = descriptor.invertYZ ? descriptor.invertYZ.valueOf() : false;
    this.alignBottom = descriptor.alignBottom ? descriptor.alignBottom.valueOf() : false;
    this.scale = descriptor.scale ? descriptor.scale.valueOf() : 1;
    if (invertYZ) { this.yIndex = 2; this.zIndex = 1; } else { this.yIndex = 1; this.zIndex = 2; }
    return this.createMeshFromText(text);
  }
  static loadMeshFromObjFile(path, descriptor) { return this.loadMeshFromObjText(fs.readFileSync(path, 'utf8'), descriptor); }
  static createMeshFromText(fileContent) {                             // obj-reader.ts:46-69
    const triangles = [];
    const lines = fileContent.split('\n');
    this.initMinMax(lines);
    for (const line of lines) {
      if (line[0] === 'v' && line[1] === ' ') this.readVertexLine(line);
      else if (line[0] === 'v' && line[1] === 't') this.readTexcoordLine(line);
      else if (line[0] === 'v' && line[1] === 'n') this.readNormalLine(line);
      else if (line[0] === 'f') this.addTriangleFromFaceData(line, triangles);
    }
    this.v = []; this.vt = []; this.vn = [];
    return triangles;
  }
  static readVertexLine(line) {
    const c = line.split(' ');
    const v = [parseFloat(c[1 + this.xIndex]), parseFloat(c[1 + this.yIndex]), parseFloat(c[1 + this.zIndex])];
    vec3.subtract(v, v, this.offsets);
    vec3.mul(v, v, [this.scale, this.scale, this.scale]);
    this.v.push(v);
  }
  static readTexcoordLine(line) { const c = line.split(' '); this.vt.push([parseFloat(c[1]), parseFloat(c[2])]); }
  static readNormalLine(line) {
    const c = line.split(' ');
    this.vn.push([parseFloat(c[1 + this.xIndex]), parseFloat(c[1 + this.yIndex]), parseFloat(c[1 + this.zIndex])]);
  }
  static addTriangleFromFaceData(line, triangles) {                    // obj-reader.ts:103-117
    line = line.replace('\n', '');
    const vd = line.split(' ');
    const triangleCount = vd.length - 3;
    for (let i = 0; i < triangleCount; ++i) {
      const triangle = new Triangle();
      triangle.color = this.color;
      this.readCorner(vd[1], triangle);
      this.readCorner(vd[this.yIndex + 1 + i], triangle);
      this.readCorner(vd[this.zIndex + 1 + i], triangle);
      triangle.calculateCentroid();
      triangles.push(triangle);
    }
  }
  static readCorner(vertexDescription, triangle) {
    const p = vertexDescription.split('/');
    triangle.corners.push(this.v[parseInt(p[0]) - 1]);
    triangle.normals.push(this.vn[parseInt(p[2]) - 1]);
    triangle.textures.push(this.vt[parseInt(p[1]) - 1]);
  }
  static initMinMax(lines) {                                           // obj-reader.ts:132-166
    for (const line of lines) {
      if (line[0] === 'v' && line[1] === ' ') {
        const c = line.split(' ');
        this.mins = [parseFloat(c[1]), parseFloat(c[2]), parseFloat(c[3])];
        this.maxs = vec3.clone(this.mins);
        break;
      }
    }
    for (const line of lines) {
      if (line[0] === 'v' && line[1] === ' ') {
        const c = line.split(' ');
        const x = parseFloat(c[1 + this.xIndex]), y = parseFloat(c[1 + this.yIndex]), z = parseFloat(c[1 + this.zIndex]);
        if (x < this.mins[this.xIndex]) this.mins[this.xIndex] = x;
        if (y < this.mins[this.yIndex]) this.mins[this.yIndex] = y;
        if (z < this.mins[this.zIndex]) this.mins[this.zIndex] = z;
        if (x > this.maxs[this.xIndex]) this.maxs[this.xIndex] = x;
        if (y > this.maxs[this.yIndex]) this.maxs[this.yIndex] = y;
        if (z > this.maxs[this.zIndex]) this.maxs[this.zIndex] = z;
      }
    }
    this.offsets = vec3.add(vec3.create(), this.mins, this.maxs);
    vec3.div(this.offsets, this.offsets, [2, 2, 2]);
    if (this.alignBottom) this.offsets[this.yIndex] = this.mins[this.yIndex];
  }
}
ObjectReader.v = []; ObjectReader.vt = []; ObjectReader.vn = [];
ObjectReader.mins = [0, 0, 0]; ObjectReader.maxs = [0, 0, 0]; ObjectReader.offsets = [0, 0, 0];
ObjectReader.alignBottom = false; ObjectReader.scale = 1;
ObjectReader.xIndex = 0; ObjectReader.yIndex = 1; ObjectReader.zIndex = 2;
module.exports = { ObjectReader };
