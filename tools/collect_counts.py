#!/usr/bin/env python3
"""What the hierarchy kernel does per frame, counted by the kernel itself: the counting builds of rt_bvh.hip
(-DRT_BVH_COUNT=<mode>: the frame's ray counter then carries one statistic) are run on the same frame as the product
build and the totals go to <out>/<KEY>__counts.json, which tools/pmc_summary.py folds into profiles/traffic.json and
bench.py prices as roofline.useful / lanes_busy.

    python tools/collect_counts.py --build                      (build container: cross-compiles tools/bin/librt355_c{1,5,6,8}.so)
    python tools/collect_counts.py C3 gpurun_out/r03            (GPU box: runs them, writes gpurun_out/r03/C3-fast-v0-n1__counts.json)

modes: 1 trips of the walk loop x 2 (a trip is four steps of every lane of a wave), 5 literal evaluations (lane),
       6 leaf tests by a lane with a live ray, 8 inner-node tests by a lane with a live ray."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = (1, 5, 6, 8)


def build():
    csrc = os.path.join(ROOT, "compute_raytracer_amd", "csrc")
    os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-function -ffp-contract=off -fno-slp-vectorize".split()
    objs = [os.path.join(csrc, o) for o in ("rt_api.o", "rt_kernels.o", "rt_triangles.o", "rt_assemble.o", "rt_comm.o")]
    for m in MODES:
        obj = "/tmp/rt_bvh_c%d.o" % m
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-DRT_BVH_COUNT=%d" % m, "-c", os.path.join(csrc, "rt_bvh.hip"), "-o", obj], check=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(ROOT, "tools", "bin", "librt355_c%d.so" % m)]
                       + objs + [obj, "-L/opt/rocm/lib", "-lrccl"], check=True)
        print("built mode", m, flush=True)


def probe(lib, cfg):
    env = dict(os.environ)
    if lib:
        env["RT355_LIB"] = lib
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "count_probe.py"), cfg], cwd=ROOT, env=env, capture_output=True, text=True, check=True)
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    if "--build" in sys.argv:
        build()
        return
    cfg, out = sys.argv[1], sys.argv[2]
    base = probe(None, cfg)
    c = {m: probe(os.path.join("tools", "bin", "librt355_c%d.so" % m), cfg)["counter"] for m in MODES}
    wave_steps = 2 * c[1]
    res = {"rays": base["counter"], "kernel_id": base["kernel_id"], "node_and_leaf_tests": c[6] + c[8], "leaf_tests": c[6], "inner_node_tests": c[8],
           "literal_tests": c[5], "wave_steps": wave_steps, "lanes_busy": (c[6] + c[8]) / (64.0 * wave_steps),
           "note": "counting builds of rt_bvh.hip (tools/collect_counts.py), one frame of %s" % cfg}
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "%s-fast-v0-n1__counts.json" % cfg)
    json.dump(res, open(path, "w"), indent=1)
    print(path, json.dumps(res))


if __name__ == "__main__":
    main()
