'use strict';
// node node/render-json-scene.js scene.json out.rgba [heatmap]
// Renders a triangle scene given as JSON in the SHAPE of the reference's SceneRaytracing object
// (camera, light, triangles[{corners,normals,textures,color}], nodes[{minCorner,maxCorner,
// leftChildIndex,primitiveCount}], blasList[{inverseModel,rootNodeIndex}], blasIndices,
// triangleIndices, tlasNodesUsed, tlasNodesMax, blasNodesUsed) through RendererRaytracing --
// i.e. through the same packing code as RR:155-230.  Used by tests/test_node_host.py.
const fs = require('fs');
const crypto = require('crypto');
const { RendererRaytracing } = require('./renderer-raytracing');

async function main() {
  const j = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
  const scene = j.scene;
  scene.update = () => {};
  const renderer = new RendererRaytracing(j.width, j.height, scene, { maxBounces: j.bounces });
  const tex = j.meshTexture ? { width: j.meshTexture.width, height: j.meshTexture.height, data: Uint8Array.from(j.meshTexture.data) } : null;
  await renderer.initialize(null, tex);
  if (process.argv[4] === 'heatmap') renderer.showHeatmap();
  await renderer.render();
  await renderer.render();      // second frame: only the per-frame buffers are re-uploaded (RR:194-195)
  const px = renderer.readPixels();
  if (process.argv[3]) fs.writeFileSync(process.argv[3], Buffer.from(px.buffer, px.byteOffset, px.byteLength));
  console.log(JSON.stringify({ rays: renderer.stats().rays, sha256: crypto.createHash('sha256').update(px).digest('hex') }));
  renderer.close();
}
main().catch((e) => { console.error(e && e.message ? e.message : e); process.exit(1); });
