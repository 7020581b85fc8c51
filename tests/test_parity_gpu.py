"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances (stated here, as BASELINE.json's north_star asks):
  * RT_MODE_STRICT: bit-exact RGBA8 and exact ray count -- tolerance 0.
  * RT_MODE_FAST  : FMA contraction changes roundings by <= a few ulp.  Away from the shader's
    discontinuities (shadow test `diff < 0.005` RK:159, `t > 0.001` HK:318, `discriminant > 0`
    HK:316, nearest-hit ties, 8-bit rounding) that moves a channel by at most 1/255.  Bound:
    >= 99.5 % of pixels within 1/255 per channel, and the ray count within 0.1 %; pixels beyond
    that are threshold flips (a shadowed/lit or hit/miss decision taken the other way), which
    the WGSL spec equally allows between two conforming GPUs.
"""
import numpy as np
import pytest

from helpers import config_inputs, diff_stats, gpu_render, oracle_render

pytestmark = pytest.mark.gpu

FAST_WITHIN1 = 0.995
FAST_RAYS_REL = 1e-3


@pytest.mark.parametrize("name", ["C1", "C2"])
def test_strict_bit_exact(oracle, name):
    cfg, scene = config_inputs(name)
    W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
    ref, _, rays = oracle_render(oracle, scene, W, H, B)
    img, st = gpu_render(scene, W, H, B, strict=True)
    d = diff_stats(img, ref)
    assert d["max"] == 0, d
    assert np.array_equal(img, ref)
    assert st["rays"] == rays


@pytest.mark.parametrize("name", ["C1", "C2"])
def test_fast_within_tolerance(oracle, name):
    cfg, scene = config_inputs(name)
    W, H, B = cfg["width"], cfg["height"], cfg["bounces"]
    ref, _, rays = oracle_render(oracle, scene, W, H, B)
    img, st = gpu_render(scene, W, H, B, strict=False)
    d = diff_stats(img, ref)
    print(name, d, st["rays"], rays)
    assert d["within1"] >= FAST_WITHIN1, d
    assert abs(st["rays"] - rays) <= FAST_RAYS_REL * rays
    assert np.all(img[..., 3] == 255)
