'use strict';
// node node/render-json-scene.js scene.json out.rgba [heatmap]
// Renders a triangle scene whose upload buffers come from somewhere else (a capture of a browser run,
// the Python mirror): scene.json = {width, height, bounces, camera:{position,forwards,right,up},
// light:{position,lightIntensity,minIntensity}, tlasNodesMax, packed:{triangleData,nodeDataB,
// triangleIndexData}, frame:{blasData,blasIndexData,nodeDataA}, meshTexture:{width,height,data}} --
// number arrays in the layouts of renderer-raytracing.ts:169-229.  Used by tests/test_node_host.py.
const fs = require('fs');
const crypto = require('crypto');
const { RendererRaytracing } = require('./renderer-raytracing');

async function main() {
  const j = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
  const f32 = (o) => { const r = {}; for (const k of Object.keys(o)) r[k] = Float32Array.from(o[k]); return r; };
  const scene = { camera: j.camera, light: j.light, spheres: [], hasTriangles: true, tlasNodesMax: j.tlasNodesMax,
                  packed: f32(j.packed), frame: f32(j.frame), update() {} };
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
