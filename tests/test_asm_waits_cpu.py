"""The hand-issued loads of rt_bvh.hip (global_load in inline asm, awaited by an s_waitcnt placed later by hand: the
candidate's exact record in the pooled evaluation, centre and colour of the sphere a reflection ray has hit) are invisible
to the compiler's own wait-count pass.  Nothing in the language stops the register allocator from copying, spilling or
reusing their destination registers between the issue and the wait -- which would read the registers before the data has
landed and give silently wrong pixels.  This test compiles the file with the product's flags (and with the
development build's) to gfx950 assembly and checks, for every such load of every kernel instantiation, that no instruction
between the load and the first hand-placed `s_waitcnt vmcnt(0)` behind it reads or writes any of its destination registers,
and that none of them goes to scratch there."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-function -ffp-contract=off -fno-slp-vectorize".split()

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(asm):
    lines = asm.split("\n")
    loads = 0
    i = 0
    in_asm = False
    while i < len(lines):
        t = lines[i].strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and t.startswith("global_load_dword"):
            dest = regs_of(t.split(",")[0])
            assert dest, t
            loads += 1
            # walk forward to the first hand-placed wait; within the same asm block several loads may follow each other
            j, inner, found = i + 1, True, False
            while j < len(lines):
                u = lines[j].strip()
                if u.startswith(";;#ASMSTART"):
                    inner = True
                elif u.startswith(";;#ASMEND"):
                    inner = False
                elif u and not u.startswith(";") and not u.startswith("."):
                    if inner and u.startswith("s_waitcnt") and "vmcnt(0)" in u:
                        found = True
                        break
                    if u.startswith(".Lfunc_end") or u.startswith("s_endpgm"):
                        break
                    touched = regs_of(u) & dest
                    # another hand-issued load into OTHER registers is fine; anything naming ours is not
                    assert not touched, "line %d: `%s` touches v%s between the hand-issued load (line %d: %s) and its wait" % (
                        j + 1, u, sorted(touched), i + 1, t)
                    assert not u.startswith("scratch_store") or not (regs_of(u) & dest), u
                j += 1
            assert found, "line %d: no hand-placed s_waitcnt vmcnt(0) behind `%s`" % (i + 1, t)
        i += 1
    return loads


@pytest.mark.parametrize("extra", [[], ["-DRT_BVH_DEV_ENV"]], ids=["product", "dev"])
def test_hand_issued_loads_are_left_alone_until_their_wait(tmp_path, extra):
    out = str(tmp_path / "rt_bvh.s")
    r = subprocess.run([HIPCC] + FLAGS + extra + ["-S", "--cuda-device-only", "-o", out,
                        os.path.join(ROOT, "compute_raytracer_amd", "csrc", "rt_bvh.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    n = check(open(out).read())
    assert n >= 30            # three loads in each of the kernel's instantiations


def test_the_check_notices_a_touched_register():
    bad = "\n".join([";;#ASMSTART", "global_load_dwordx4 v[4:7], v1, s[2:3]", ";;#ASMEND", "v_mov_b32_e32 v9, v5",
                     ";;#ASMSTART", "s_waitcnt vmcnt(0)", ";;#ASMEND"])
    with pytest.raises(AssertionError):
        check(bad)
    good = bad.replace("v9, v5", "v9, v8")
    assert check(good) == 1
