'use strict';
// node node/build-obj-scene.js spec.json [out.rgba]
// spec: {width,height,bounces,updates:[dt,...], meshes:[{obj:"<OBJ text>",descriptor:{color,alignBottom,invertYZ,scale}}],
//        models:[{meshIndex,position,eulers,eulerSpeed}], meshTexture:{width,height,data}}
// Builds the triangle scene ENTIRELY in JS (OBJ text -> soup -> SAH tree -> instances -> top-level tree:
// the reference's createScene path), prints the f32 bit patterns of the upload buffers (so that the
// Python mirror can be compared bit for bit) and, when a GPU is present and out.rgba is given, renders
// through the addon.
const fs = require('fs');
const crypto = require('crypto');
const { SceneRaytracing, loadMesh, makeModel } = require('./scene-raytracing');

async function main() {
  const spec = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
  const scene = new SceneRaytracing();
  await scene.createScene([]);
  const meshes = spec.meshes.map((m) => loadMesh(m.obj, m.descriptor));
  const models = spec.models.map((m) => makeModel(m.meshIndex, m.position, m.eulers, m.eulerSpeed));
  scene.createTriangleScene(meshes, models);
  for (const dt of spec.updates || []) scene.update(dt);
  const bits = (a) => Array.from(new Uint32Array(a.buffer, a.byteOffset, a.length));
  const sha = (a) => crypto.createHash('sha256').update(Buffer.from(a.buffer, a.byteOffset, a.byteLength)).digest('hex');
  const out = {
    nTriangles: scene.triangleCount, tlasNodesUsed: scene.tlasNodesUsed, tlasNodesMax: scene.tlasNodesMax,
    blasNodesUsed: scene.blasNodesUsed,
    blas: bits(scene.frame.blasData), blasIndices: bits(scene.frame.blasIndexData), tlasNodes: bits(scene.frame.nodeDataA),
    blasNodes: bits(scene.packed.nodeDataB), triangleIndices: bits(scene.packed.triangleIndexData),
    triangles: bits(scene.packed.triangleData), centroids0: bits(meshes[0].soup.centroid),
    sha: { triangles: sha(scene.packed.triangleData), blasNodes: sha(scene.packed.nodeDataB), lookup: sha(scene.packed.triangleIndexData),
           blas: sha(scene.frame.blasData), tlasNodes: sha(scene.frame.nodeDataA) },
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
