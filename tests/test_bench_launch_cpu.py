"""bench.py's front door for N > 1: `python bench.py --gpus N` with no launcher around it starts the ranks itself, as a child
`python -m torch.distributed.run ...`, before the parent has imported torch or touched HIP (a process that has initialised the
GPU must never be replaced, and the parent makes no GPU call at all); it relays the child's output and exit code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_dry_launch_prints_the_child_command():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "7", "--warmup", "1", "--dry-launch"],
                         capture_output=True, text=True, env=clean_env(), timeout=60)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "2", "--steps", "7", "--warmup", "1"]       # the same arguments, without --dry-launch


def test_the_parent_touches_neither_torch_nor_hip():
    code = ("import sys, runpy; sys.argv = [%r, '--gpus', '2', '--dry-launch']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert not e.code, e.code\n"
            "bad = [m for m in sys.modules if m == 'torch' or m.startswith('torch.') or m.startswith('compute_raytracer_amd')]\n"
            "print('LOADED', bad)\n" % (BENCH, BENCH))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=clean_env(), timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1] == "LOADED []"


def test_under_a_launcher_nothing_is_launched():
    """WORLD_SIZE in the environment = a launcher started this rank: no child, the rank runs (and, here, finds no GPU)."""
    env = dict(clean_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, env=env, timeout=300)
    assert "launch" not in out.stdout
    assert out.returncode != 0 and "no GPU visible" in out.stderr


def test_the_ranks_run_as_children_and_their_exit_code_comes_back():
    """No GPU here: both ranks stop with "no GPU visible; the hot path has no CPU fallback", torch.distributed.run reports the
    failure, and the parent's exit code is the child's (non-zero), with nothing on stdout."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=clean_env(), timeout=600)
    assert out.returncode != 0
    assert "no GPU visible" in out.stderr
    assert out.stdout.strip() == ""
