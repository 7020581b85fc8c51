'use strict';
// node node/bench-frames.js C3 [frames]            -- BASELINE config, the scene from this host layer's own generator
// node node/bench-frames.js scene.json [frames]    -- a packed triangle scene (render-json-scene.js's format, plus
//                                                     sky: {size, file}: six size x size rgba8 faces, raw, +X -X +Y -Y +Z -Z)
// The drop-in measured where the north_star puts it -- in the reference's host language, on the headline (VERDICT r04, item 3):
//   awaitedMsPerFrame   the reference's loop (src/app.ts:117-128): camera.move, `await renderer.render()` (recalculateScene,
//                       rt_render, rt_wait in a worker thread: RR:435-469), one frame at a time;
//   inflightMsPerFrame  the same frames WITHOUT the await: recalculateScene + rt.render back to back (the library keeps four
//                       on the device), one `await rt.wait` per batch;
//   streamedMsPerFrame  ... with every frame copied to pinned host memory by readPixelsAsync two frames behind.
// Prints one JSON line (sha256 of the last frame: bench.py holds it against the oracle's).
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const { SceneRaytracing, syntheticSpheres, BASELINE_CONFIGS } = require('./scene-raytracing');
const { RendererRaytracing } = require('./renderer-raytracing');
const { CubemapMaterial } = require('./cubemap-material');
const rt = require(path.join(__dirname, 'rt355.node'));

const now = () => Number(process.hrtime.bigint()) / 1e6;
const median = (a) => a.slice().sort((x, y) => x - y)[a.length >> 1];

async function main() {
  const what = process.argv[2] || 'C3';
  const frames = parseInt(process.argv[3] || '100', 10);
  let scene, width, height, bounces, sky = null, tex = null;
  if (BASELINE_CONFIGS[what]) {
    const cfg = BASELINE_CONFIGS[what];
    scene = new SceneRaytracing();
    await scene.createScene(syntheticSpheres(cfg.spheres, cfg.seed));
    width = cfg.width; height = cfg.height; bounces = cfg.bounces;
  } else {
    const j = JSON.parse(fs.readFileSync(what, 'utf8'));
    const f32 = (o) => { const r = {}; for (const k of Object.keys(o)) r[k] = Float32Array.from(o[k]); return r; };
    scene = { camera: Object.assign({ move() {} }, j.camera), light: j.light, spheres: [], hasTriangles: true, tlasNodesMax: j.tlasNodesMax,
              packed: f32(j.packed), frame: f32(j.frame), update() {} };
    width = j.width; height = j.height; bounces = j.bounces;
    if (j.meshTexture) tex = { width: j.meshTexture.width, height: j.meshTexture.height, data: Uint8Array.from(j.meshTexture.data) };
    if (j.sky) {
      const raw = fs.readFileSync(j.sky.file), n = j.sky.size;
      sky = new CubemapMaterial();
      for (let f = 0; f < 6; ++f) sky.faces.push({ width: n, height: n, data: new Uint8Array(raw.buffer, raw.byteOffset + f * n * n * 4, n * n * 4) });
    }
  }
  const renderer = new RendererRaytracing(width, height, scene, { maxBounces: bounces });
  await renderer.initialize(sky, tex);
  const ctx = renderer.ctx;
  for (let i = 0; i < 8; ++i) await renderer.render();

  const reps = 5;
  const awaited = [];
  for (let r = 0; r < reps; ++r) {
    const t = now();
    for (let i = 0; i < frames; ++i) { scene.camera.move(0, 0); await renderer.render(); }
    awaited.push((now() - t) / frames);
  }
  const kernelMs = renderer.stats().kernelMs, kernelIdAwaited = renderer.stats().kernelId;

  const batch = async (n, copyTo) => {
    let done = 0;
    while (done < n) {
      const chunk = Math.min(n - done, 48);                    // the library's event ring holds 64 frames
      for (let i = 0; i < chunk; ++i) {
        scene.camera.move(0, 0);
        renderer.recalculateScene();
        rt.render(ctx);
        if (copyTo && done + i >= 2) rt.readPixelsAsync(ctx, 2, copyTo[(done + i) % 4]);
      }
      done += chunk;
      await rt.wait(ctx);
    }
    if (copyTo) { rt.readPixelsAsync(ctx, 1, copyTo[(n + 2) % 4]); rt.readPixelsAsync(ctx, 0, copyTo[(n + 3) % 4]); rt.readPixelsWait(ctx); }
  };
  await batch(16, null);
  const inflight = [];
  for (let r = 0; r < reps; ++r) { const t = now(); await batch(frames, null); inflight.push((now() - t) / frames); }
  const kernelIdInflight = renderer.stats().kernelId;
  const host = [];
  for (let i = 0; i < 4; ++i) host.push(rt.hostAlloc(width * height * 4));
  await batch(8, host);
  const streamed = [];
  for (let r = 0; r < 3; ++r) { const t = now(); await batch(Math.max(24, frames >> 1), host); streamed.push((now() - t) / Math.max(24, frames >> 1)); }

  await renderer.render();
  const px = renderer.readPixels();
  const st = renderer.stats();
  console.log(JSON.stringify({ what, width, height, frames, repeats: reps, rays: st.rays,
    awaitedMsPerFrame: median(awaited), awaitedMsPerFrameMin: Math.min.apply(null, awaited),
    inflightMsPerFrame: median(inflight), inflightMsPerFrameMin: Math.min.apply(null, inflight),
    streamedMsPerFrame: median(streamed), kernelMs, kernelIdAwaited, kernelIdInflight,
    hwQueuesEnv: process.env.GPU_MAX_HW_QUEUES || null, buildId: rt.buildId(),
    sha256: crypto.createHash('sha256').update(px).digest('hex') }));
  renderer.close();
}
main().catch((e) => { console.error(e && e.stack ? e.stack : e); process.exit(1); });
