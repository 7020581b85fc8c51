'use strict';
// node node/build-obj-scene.js spec.json [out.rgba]
// spec: {width,height,bounces,updates:[dt,...], meshes:[{obj:"<OBJ text>",descriptor:{color,alignBottom,invertYZ,scale}}],
//        models:[{meshIndex,position,eulers,eulerSpeed}], meshTexture:{width,height,data}}
// Builds the triangle scene ENTIRELY in JS (OBJ reader -> SAH BVH -> models -> TLAS, the reference's
// createScene path), prints the packed buffers' f32 bit patterns (so that the Python mirror can be
// compared bit for bit) and, when a GPU is present and out.rgba is given, renders through the addon.
const fs = require('fs');
const crypto = require('crypto');
const { SceneRaytracing } = require('./scene-raytracing');
const { Mesh } = require('./mesh');
const { Model } = require('./model/model');

async function main() {
  const spec = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
  const scene = new SceneRaytracing();
  await scene.createScene([]);
  const meshes = spec.meshes.map((m) => new Mesh().initializeFromText(m.obj, m.descriptor));
  const models = spec.models.map((m) => new Model(m.meshIndex, m.position, m.eulers, m.eulerSpeed));
  scene.createTriangleScene(meshes, models);
  for (const dt of spec.updates || []) scene.update(dt);
  const bits = (a) => Array.from(new Uint32Array(Float32Array.from(a).buffer));
  const out = {
    nTriangles: scene.triangles.length, tlasNodesUsed: scene.tlasNodesUsed, tlasNodesMax: scene.tlasNodesMax,
    blasNodesUsed: scene.blasNodesUsed, blasIndices: scene.blasIndices, triangleIndices: scene.triangleIndices,
    blas: scene.blasList.map((b) => bits(Array.from(b.inverseModel).concat([b.rootNodeIndex]))),
    nodes: scene.nodes.map((n) => bits([n.minCorner[0], n.minCorner[1], n.minCorner[2], n.leftChildIndex,
                                        n.maxCorner[0], n.maxCorner[1], n.maxCorner[2], n.primitiveCount])),
    tri0: bits([].concat(scene.triangles[0].corners[0], scene.triangles[0].corners[1], scene.triangles[0].corners[2],
                         Array.from(scene.triangles[0].centroid))),
  };
  if (process.argv[3]) {
    const { RendererRaytracing } = require('./renderer-raytracing');
    const renderer = new RendererRaytracing(spec.width, spec.height, scene, { maxBounces: spec.bounces });
    const tex = spec.meshTexture ? { width: spec.meshTexture.width, height: spec.meshTexture.height, data: Uint8Array.from(spec.meshTexture.data) } : null;
    await renderer.initialize(null, tex);
    await renderer.render();
    const px = renderer.readPixels();
    fs.writeFileSync(process.argv[3], Buffer.from(px.buffer, px.byteOffset, px.byteLength));
    out.rays = renderer.stats().rays;
    out.sha256 = crypto.createHash('sha256').update(px).digest('hex');
    renderer.close();
  }
  console.log(JSON.stringify(out));
}
main().catch((e) => { console.error(e && e.stack ? e.stack : e); process.exit(1); });
