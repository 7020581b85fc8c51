'use strict';
// node node/bench-loop.js spec.json [frames]
// The reference's animation loop (src/app.ts:117-128) through the addon, in the reference's host language: per frame
// scene.update(dt) -- models spin, instance matrices and the top-level tree are rebuilt (scene-raytracing.ts:138-143) --,
// camera.move, `await renderer.render()` (recalculateScene: the params write and the three per-frame instance writes; rt_render;
// rt_wait in a worker).  Timed against the same loop WITHOUT scene.update (a static scene), both one frame at a time.
// spec: as node/build-obj-scene.js.  Prints one JSON line.
const fs = require('fs');
const { SceneRaytracing, loadMesh, makeModel } = require('./scene-raytracing');
const { RendererRaytracing } = require('./renderer-raytracing');

async function main() {
  const spec = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
  const frames = parseInt(process.argv[3] || '100', 10);
  const t0 = Date.now();
  const scene = new SceneRaytracing();
  await scene.createScene([]);
  const meshes = spec.meshes.map((m) => loadMesh(m.obj, m.descriptor));
  const models = spec.models.map((m) => makeModel(m.meshIndex, m.position, m.eulers, m.eulerSpeed));
  scene.createTriangleScene(meshes, models);
  const buildMs = Date.now() - t0;
  const renderer = new RendererRaytracing(spec.width, spec.height, scene, { maxBounces: spec.bounces });
  const tex = spec.meshTexture ? { width: spec.meshTexture.width, height: spec.meshTexture.height, data: Uint8Array.from(spec.meshTexture.data) } : null;
  await renderer.initialize(null, tex);
  const now = () => Number(process.hrtime.bigint()) / 1e6;
  for (let i = 0; i < 8; ++i) await renderer.render();
  let t = now();
  for (let i = 0; i < frames; ++i) { scene.camera.move(0, 0); await renderer.render(); }
  const staticMs = (now() - t) / frames;
  let hostMs = 0;
  t = now();
  for (let i = 0; i < frames; ++i) {
    const h = now();
    scene.update(0.016);
    scene.camera.move(0, 0);
    hostMs += now() - h;
    await renderer.render();
  }
  const animatedMs = (now() - t) / frames;
  const st = renderer.stats();
  console.log(JSON.stringify({ triangles: scene.triangleCount, width: spec.width, height: spec.height, frames, sceneBuildMs: buildMs,
    staticLoopMsPerFrame: staticMs, animatedLoopMsPerFrame: animatedMs, ratio: animatedMs / staticMs,
    hostSceneUpdateMsPerFrame: hostMs / frames, kernelMs: st.kernelMs, kernelId: st.kernelId, rays: st.rays }));
  renderer.close();
}
main().catch((e) => { console.error(e && e.message ? e.message : e); process.exit(1); });
