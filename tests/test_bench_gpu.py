"""bench.py and __graft_entry__.smoke() on the GPU box: the JSON contract, and the N>1 code path
(rt_comm_init + rt_render_gather: render -> RCCL exchange -> de-interleave inside librt355.so)
taken with a single rank, its assembled frame hashed against the oracle's golden frame."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), cwd=ROOT, env=e,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    # the contract: ONE line on stdout (native banners of gloo / RCCL included -- bench.py keeps them on stderr)
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[:2000]
    return json.loads(lines[0]), out.stderr


def test_bench_json_contract():
    d, _ = run_bench("--steps", "3", "--warmup", "1", "--config", "C2", "--cpu-seconds", "2")
    for k in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "serial_ms_per_step", "frame_check"]:
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["config"]["rays_per_frame"] == 9061272                      # C2, the oracle's count
    r = d["roofline"]
    assert r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert r["hbm"]["bytes_per_launch"] == 4 * 1920 * 1080 + 32 * 64 + 96
    assert d["frame_check"]["sha256_matches_oracle_frame"] is True
    assert d["serial_ms_per_step"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "tiles" in c["sample"]
    assert abs(d["value"] - d["config"]["rays_per_frame"] / d["ms_per_step"] / 1e3) / d["value"] < 0.02


def test_bench_headline_roofline_is_an_executed_fraction_below_one():
    """roofline.frac prices what the kernel executes (PMC instruction count from the
    committed profile) and is <= 1; the algorithmic brute-force rate is reported apart."""
    d, _ = run_bench("--steps", "8", "--warmup", "2", "--no-cpu-baseline")
    fr = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["C3"]
    assert d["config"]["rays_per_frame"] == fr["rays"] and d["frame_check"]["sha256_matches_oracle_frame"] is True
    r = d["roofline"]
    if r["frac"] is None:
        # profiles are evidence for the build they were taken with (rt_build_id): a library built from other sources
        # reports no executed-instruction fraction instead of a stale one
        assert "re-run tools/collect_profiles.sh" in r["basis"] and d["kernel"]["build_id"] in r["basis"]
        assert "bvh_pixels" in r["kernel"] and r["launches_in_flight"] == 4
        pytest.skip("profiles/traffic.json is stale for this build: " + r["basis"])
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["valu_wave_insts_per_launch"] * 128 / (r["time_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert r["from_profile"]["key"] == "C3/fast/v0/n1" and r["from_profile"]["files"]
    assert 0.05 < r["serial"]["frac"] <= r["frac"] * 1.05 and d["serial_ms_per_step"] >= d["ms_per_step"] * 0.9
    assert d["algorithmic_speedup_vs_bruteforce"]["value"] > 1.0
    assert r["launches_in_flight"] == 4 and "bvh_pixels" in r["kernel"]
    # verdict r2 item 4: the kernel is the one the library says it launched; what the issue slots are used FOR is reported
    # beside the issue fraction; the line leads with what a caller gets
    assert d["kernel"]["id"] == 4 and d["kernel"]["name"] == "bvh_pixels<8>" and len(d["kernel"]["build_id"]) == 16
    u = r["useful"]
    assert 0.0 < u["frac"] < r["frac"] and 0.3 < u["lanes_busy"] < 1.0
    assert 30 < u["node_and_leaf_tests_per_ray"] < 70 and 1 < u["literal_tests_per_ray"] < 5
    assert d["serial_ms_per_step"] <= d["readback_ms_per_step"] and abs(d["serial_value"] - d["config"]["rays_per_frame"] / d["serial_ms_per_step"] / 1e3) < 1.0
    # round 5: the K steps are timed `repeats` times in one invocation (the first region is ms_per_step), the clock state is on the
    # line, and the same workload has been measured through the Node host in a child process
    assert d["repeats"] == 5 and len(d["ms_per_step_all"]) == 5 and d["ms_per_step_all"][0] == d["ms_per_step"]
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= max(d["ms_per_step_all"])
    assert max(d["ms_per_step_all"]) < 1.5 * d["ms_per_step_min"]
    assert d["serial_ms_per_step_min"] <= d["serial_ms_per_step_median"]
    assert d["clocks"] is None or d["clocks"]["values"]
    n = d["node"]
    if "skipped" not in n:
        assert n["frame_matches_oracle"] is True and n["rays"] == fr["rays"]
        assert 0.7 < d["node_loop_ms_per_step"] / d["serial_ms_per_step_median"] < 1.4
        assert 0.7 < d["node_inflight_ms_per_step"] / d["ms_per_step_median"] < 1.4


def test_bench_distributed_path_with_one_rank():
    fr = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["C2"]
    for gather in ("root", "all"):
        d, err = run_bench("--steps", "2", "--warmup", "1", "--config", "C2", "--no-cpu-baseline", "--force-dist",
                           "--gather", gather, env={"MASTER_PORT": "29541"})
        assert d["frame_check"]["sha256_matches_oracle_frame"] is True, gather
        assert d["config"]["rays_per_frame"] == fr["rays"] and "cpu_baseline" not in d
        assert "rt_render_gather" in d["config"]["parallelism"] and "gather_ms_avg" in d


def test_bench_two_ranks_rehearsed_on_one_device():
    """The driver's N > 1 command on a one-GPU box: `bench.py --gpus 2` starts its two ranks itself (a child
    torch.distributed.run), RT355_BENCH_ONE_DEVICE puts both on device 0.  RCCL refuses two ranks of one device: every rank
    falls back TOGETHER to the labelled host exchange (rows read back, all-gathered over gloo), and the frame the two ranks'
    interleaved tiles make -- 135 tiles: 68 and 67, ragged -- is the oracle's.  (Should RCCL ever accept it: the RCCL path, same frame.)"""
    d, err = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "C2", "--no-cpu-baseline", "--repeats", "1",
                       env={"RT355_BENCH_ONE_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["exchange"] in ("host", "rccl")
    if d["exchange"] == "host":
        assert "rt_comm_init" in d["config"]["parallelism"] and "FALLBACK" in d["config"]["parallelism"]
        assert "ncclCommInitRank" in d["exchange_error"]
    assert d["frame_check"]["sha256_matches_oracle_frame"] is True
    assert d["config"]["rays_per_frame"] == json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["C2"]["rays"]


def test_smoke_entry_point():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "smoke ok" in out.stdout, out.stderr[-2000:]


def test_bench_distributed_path_pipelines_hierarchy_frames():
    """C3 through the N>1 code path with one rank: 7 frames rotate over the four streams and buffer
    sets (render -> RCCL exchange -> assemble, all inside rt_render_gather), the assembled frame is the
    golden one."""
    fr = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["C3"]
    d, err = run_bench("--steps", "6", "--warmup", "1", "--config", "C3", "--no-cpu-baseline", "--force-dist",
                       env={"MASTER_PORT": "29542"})
    assert d["frame_check"]["sha256_matches_oracle_frame"] is True
    assert d["config"]["rays_per_frame"] == fr["rays"]
    assert d["roofline"]["launches_in_flight"] == 4 and "bvh_pixels" in d["roofline"]["kernel"]


def test_bench_serial_mode_times_single_launches():
    d, _ = run_bench("--steps", "4", "--warmup", "1", "--config", "C3", "--no-cpu-baseline", "--serial")
    r = d["roofline"]
    assert r["launches_in_flight"] == 1 and d["config"]["frames_in_flight"] == 1
    assert abs(r["time_ms"] - r["kernel_ms_avg"]) < 1e-9
    assert r["kernel_ms_avg"] <= d["ms_per_step"] * 1.05          # a launch is the bulk of a serial step
    assert abs(d["serial_ms_per_step"] - d["ms_per_step"]) < 1e-9


def test_bench_reference_scene_config():
    """`--config REF`: the reference's own scene in the state of its screenshot (tests/golden/ref_scene.npz), the window and
    bounce count its overlay reports "6 ms" for; the timed frame is the oracle's frame, which is the screenshot's outside
    the missing texture; an animated frame (instances rewritten every frame, RR:169-192) is reported beside the static one."""
    d, _ = run_bench("--steps", "12", "--warmup", "2", "--config", "REF", "--cpu-seconds", "3")
    assert "12604 triangles" in d["metric"] and d["config"]["workload"].startswith("REF: 1344x846")
    assert d["frame_check"]["sha256_matches_oracle_frame"] is True and d["frame_check"]["sampled_tiles_match_oracle"] is True
    assert d["kernel"]["name"] == "trace_triangles"
    assert 0 < d["ms_per_step"] <= d["serial_ms_per_step"] * 1.05 < 6.0           # against the reference's 6 ms on its unnamed GPU
    assert d["animated_ms_per_step"] >= d["serial_ms_per_step"] * 0.8 and d["animated_host_scene_update_ms"] > 0
    assert d["cpu_baseline"]["kind"] == "port"
    if "skipped" not in d["node"]:                  # the Node host on the packed buffers of the same scene: the same frame
        assert d["node"]["frame_matches_oracle"] is True and d["node_loop_ms_per_step"] > 0 and d["node_inflight_ms_per_step"] > 0


def test_bench_triangle_config_prices_gathers_against_l2():
    """the reference's live scene type gets the same evidence as the sphere kernel:
    `--config TRI` (12.8 k procedural triangles, the reference screenshot's window, 4 bounces) with a
    bytes-based roofline (32-B node + 160-B triangle + 80-B instance gathers per ray, counted by the
    oracle on the sampled tiles, against the L2 roof) and the GPU's sampled rows checked against the oracle."""
    d, _ = run_bench("--steps", "6", "--warmup", "2", "--config", "TRI", "--cpu-seconds", "4")
    assert "triangles" in d["metric"] and d["config"]["rays_per_frame"] > 1344 * 846
    r = d["roofline"]
    assert r["bound"].startswith("latency") and r["unit"] == "GB/s" and r["peak"] == 34500.0
    if r["frac"] is not None:      # measured L2 read traffic (TCP / TCC passes of this build): a small fraction of the L2 roof
        assert 0.0 < r["frac"] < 0.2 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert 0.3 < r["l2"]["hit_rate"] <= 1.0 and r["l2"]["mean_read_latency_cycles"] > 50
    else:
        assert "re-run tools/collect_profiles.sh" in r["basis"] or "no TCP / TCC pass" in r["basis"]
    q = r["requested"]
    assert q["gathers_per_ray"]["node_loads_32B"] > 5 and q["gathers_per_ray"]["triangle_tests_48B"] > 0.5 and q["frac"] > 0
    assert d["frame_check"]["sampled_tiles_match_oracle"] is True
    assert d["cpu_baseline"]["kind"] == "port" and "triangle path" in d["cpu_baseline"]["sample"]
