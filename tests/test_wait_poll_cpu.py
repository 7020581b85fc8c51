"""rt_wait on a context with an RCCL communicator polls instead of blocking: the frames' end events, the
communicator's asynchronous error state and the caller's deadline (rt_set_comm_timeout).  The decision
logic lives in a header without HIP or RCCL in it (csrc/rt_wait_poll.h) and is exercised here with stub
environments; the GPU side (a communicator of one still completes, the deadline API) is in
tests/test_c_abi_gpu.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_wait_poll_state_machine(tmp_path):
    exe = str(tmp_path / "wait_poll_test")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c", "wait_poll_test.cpp"), "-o", exe],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "wait poll ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_comm_entry_points_poison_on_failure():
    """Source-level guard: every failure between the render's enqueue and the end event poisons the
    communicator, and a poisoned communicator refuses collectives."""
    src = open(os.path.join(ROOT, "compute_raytracer_amd", "csrc", "rt_comm.hip")).read()
    body = src[src.index("int rt_render_gather(rt_ctx* c, int root)"):src.index("int rt_set_comm_timeout")]
    assert body.count("poison(c,") >= 5 and "RT_NCCL(" not in body
    group = src[src.index("int rt_group_render(rt_group* g, int root)"):src.index("int rt_group_wait")]
    assert group.count("poison_all(") >= 7 and "RT_NCCL(" not in group and "RT_HIP(" not in group
    assert "if (c->comm->poisoned) return fail(RT_ERR_COMM" in src
    assert "ncclCommAbort" in src and "ncclCommGetAsyncError" in src
