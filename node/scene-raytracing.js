'use strict';
// SceneRaytracing for sphere scenes -- the counterpart of src/rendering-raycast/scene-raytracing.ts
// (`spheres: Sphere[]` is declared there, line 18, but never filled).  Same deterministic
// generator as compute_raytracer_amd/scene_raytracing.py (SURVEY.md 8(d)); BigInt splitmix64.
const { Camera } = require('./camera');
const { Sphere } = require('./sphere');

const M64 = (1n << 64n) - 1n;
const BASELINE_CONFIGS = {
  C1: { width: 256, height: 256, spheres: 3, bounces: 1, seed: 356 },
  C2: { width: 1920, height: 1080, spheres: 64, bounces: 4, seed: 357 },
  C3: { width: 3840, height: 2160, spheres: 1024, bounces: 8, seed: 358 },
  C4: { width: 3840, height: 2160, spheres: 1024, bounces: 8, seed: 358 },
};
const CONSTANT_SKY_RGBA = [128, 179, 255, 255];

class SplitMix64 {
  constructor(seed) { this.state = BigInt(seed) & M64; }
  nextU64() {
    this.state = (this.state + 0x9E3779B97F4A7C15n) & M64;
    let z = this.state;
    z = ((z ^ (z >> 30n)) * 0xBF58476D1CE4E5B9n) & M64;
    z = ((z ^ (z >> 27n)) * 0x94D049BB133111EBn) & M64;
    return z ^ (z >> 31n);
  }
  uniform() { return Number(this.nextU64() >> 40n) / 16777216.0; }
  range(lo, hi) { return lo + (hi - lo) * this.uniform(); }
}

function syntheticSpheres(n, seed) {
  if (n < 1) return [];
  const rng = new SplitMix64(seed);
  const spheres = [new Sphere([0.0, -100.0, 0.0], 100.0, [0.8, 0.8, 0.8])];
  const rscale = Math.fround(Math.pow(64.0 / n, 1.0 / 3.0));
  for (let i = 0; i < n - 1; ++i) {
    const x = rng.range(-12.0, 12.0);
    const z = rng.range(-26.0, -3.0);
    let r = rng.range(0.5, 1.5) * rscale;
    r = Math.min(Math.max(r, 0.04), 1.5);
    const y = r + rng.range(0.0, 3.0);
    const col = [rng.range(0.2, 1.0), rng.range(0.2, 1.0), rng.range(0.2, 1.0)];
    spheres.push(new Sphere([x, y, z], r, col));
  }
  return spheres;
}

class SceneRaytracing {
  constructor() { this.camera = null; this.light = null; this.spheres = []; }
  async createScene(spheres) {                                       // scene-raytracing.ts:37-45
    this.camera = new Camera([0.0593, 2.692, 3.293], 106, 270);
    this.light = { position: [0, 5, 0], lightIntensity: 3.0, minIntensity: 0.3 };
    this.spheres = spheres || [];
    return this;
  }
  update(dt) { return dt; }                                          // scene-raytracing.ts:138-143: static spheres
}
module.exports = { SceneRaytracing, syntheticSpheres, SplitMix64, BASELINE_CONFIGS, CONSTANT_SKY_RGBA };
