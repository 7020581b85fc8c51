"""Value-identical shortcuts of the device code (rt_device.h), checked where a check can be exhaustive: on the CPU, in
IEEE fp32 (numpy's float32 square root is correctly rounded, as the device's sqrtf under -fno-fast-math is)."""
import numpy as np

f32 = np.float32


def test_diff_threshold_on_the_squared_length():
    """light_term: `length(v) < 0.005` decided as `dot(v, v) < 0x1.a36e2cp-16`: the threshold is the smallest float whose
    correctly rounded root reaches 0.005f, and the root is monotone."""
    t = f32(float.fromhex("0x1.a36e2cp-16"))
    below = np.nextafter(t, f32(0), dtype=f32)
    assert np.sqrt(t, dtype=f32) >= f32(0.005) and np.sqrt(below, dtype=f32) < f32(0.005)
    bits = np.arange(int(t.view(np.uint32)) - 3_000_000, int(t.view(np.uint32)) + 3_000_000, dtype=np.uint32)
    x = bits.view(f32)
    assert np.array_equal(np.sqrt(x, dtype=f32) < f32(0.005), x < t)
    for v in [f32(0), f32(1e-30), f32(1), f32(np.inf), f32(np.nan)]:
        with np.errstate(invalid="ignore"):
            assert bool(np.sqrt(v, dtype=f32) < f32(0.005)) == bool(v < t)


def test_root_of_a_number_next_to_one():
    """length_of_unit: for x = 1 + k ulps, |k| <= 4096, the correctly rounded root has the bits 0x3F800000 + (k >> 1)."""
    k = np.arange(-4096, 4097, dtype=np.int64)
    x = (0x3F800000 + k).astype(np.uint32).view(f32)
    want = np.sqrt(x, dtype=f32).view(np.uint32).astype(np.int64)
    assert np.array_equal(want, 0x3F800000 + (k >> 1))


def test_flat_sky_term_is_zero_or_nan_like_the_quotients():
    """bvh_pixels, FLAT: (sc / ma) * 0 + (tc / ma) * 0 of cube_sample<1> against (r.x * 0 + r.y * 0) + r.z * 0 for directions
    that are outputs of normalize(): both are (+-)0 for finite directions and NaN when a component is."""
    rng = np.random.default_rng(3)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore", under="ignore"):
        for _ in range(20000):
            # what the kernel normalises: differences of scene coordinates (fast mode: below 2^21) and sums of three unit
            # vectors -- never long enough for the squared length to overflow (the one case in which normalize() yields
            # the zero vector, which the quotients turn into NaN and the products do not); arbitrarily short, NaN, inf
            v = rng.normal(size=3).astype(f32) * f32(10.0 ** rng.uniform(-30, 18))
            if rng.random() < 0.2:
                v[int(rng.integers(0, 3))] = f32(rng.choice([0.0, np.nan, np.inf, -np.inf, 1e-45, 1e18]))
            ln = np.sqrt(f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2])), dtype=f32)
            r = (v / ln).astype(f32)                                   # normalize(): RK:130, 147, HK:320, ray generation
            ax, ay, az = np.abs(r)
            if az >= ax and az >= ay: sc, tc, ma = r[0], r[1], az
            elif ay >= ax:            sc, tc, ma = r[0], r[2], ay
            else:                     sc, tc, ma = r[2], r[1], ax
            ref = f32(f32(f32(sc / ma) * f32(0)) + f32(f32(tc / ma) * f32(0)))
            mine = f32(f32(f32(r[0] * f32(0)) + f32(r[1] * f32(0))) + f32(r[2] * f32(0)))
            assert np.isnan(ref) == np.isnan(mine), (v, r)
            if not np.isnan(ref):
                assert ref == 0 and mine == 0
