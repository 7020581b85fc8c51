'use strict';
// Headless counterpart of src/app.ts: the call sequence of App.initialize()/_run() (app.ts:56-60,
// 117-128) -- createScene, new RendererRaytracing, initialize, then per frame scene.update,
// camera.move, await renderer.render -- for one BASELINE config; writes the RGBA8 frame to a file
// and prints one JSON line.   node node/app.js C1 out.rgba [frames] [strict]
const fs = require('fs');
const crypto = require('crypto');
const { SceneRaytracing, syntheticSpheres, BASELINE_CONFIGS } = require('./scene-raytracing');
const { RendererRaytracing } = require('./renderer-raytracing');

async function main() {
  const name = process.argv[2] || 'C1';
  const out = process.argv[3];
  const frames = parseInt(process.argv[4] || '1', 10);
  const strict = process.argv[5] === 'strict';
  const cfg = BASELINE_CONFIGS[name];
  if (!cfg) throw new Error('unknown config ' + name);
  const scene = new SceneRaytracing();
  await scene.createScene(syntheticSpheres(cfg.spheres, cfg.seed));
  const opts = { maxBounces: cfg.bounces };
  if (process.argv[6] === 'group') opts.devices = 0;
  const renderer = new RendererRaytracing(cfg.width, cfg.height, scene, opts);
  await renderer.initialize();
  renderer.setStrict(strict);
  for (let i = 0; i < frames; ++i) {
    scene.update(0);
    scene.camera.move(0, 0);
    await renderer.render();
  }
  const px = renderer.readPixels();
  const st = renderer.stats();
  if (out) fs.writeFileSync(out, Buffer.from(px.buffer, px.byteOffset, px.byteLength));
  console.log(JSON.stringify({ config: name, width: cfg.width, height: cfg.height, rays: st.rays, kernelMs: st.kernelMs,
    frames: st.frames, renderTimeMs: renderer.renderTime, sha256: crypto.createHash('sha256').update(px).digest('hex') }));
  renderer.close();
}
main().catch((e) => { console.error(e && e.message ? e.message : e); process.exit(1); });
