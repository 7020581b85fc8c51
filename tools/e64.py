"""Post-pass over the compiler's gfx950 assembly: re-encode the VOP2 (32-bit) forms of the plain arithmetic
instructions as VOP3 (64-bit).  Same instructions, same operands, same results -- only the encoding changes.
On MI355X the VOP3 encodings of v_add/v_sub/v_mul (f32), v_add/v_sub (u32) and v_and/v_or/v_xor issue in
~2.5 cycles per wave-instruction where the VOP2 encodings take ~3.4 (tools/ubench2.hip); v_fmac_f32 becomes
the equivalent v_fma_f32 with the destination as addend.  Instructions with a 32-bit literal operand stay as they
are (VOP3 has no literal on gfx9), and so do SDWA / DPP forms.
usage: python tools/e64.py in.s out.s"""
import re
import sys

PLAIN = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
         "v_and_b32", "v_or_b32", "v_xor_b32"}
INLINE_F = {"0", "0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0", "0.15915494", "0.15915494309189532"}
OPERAND = re.compile(r"^(v\d+|s\d+|vcc_lo|vcc_hi|m0|exec_lo|exec_hi|-?\d+(\.\d+)?|0x[0-9a-fA-F]+|[-|a-z0-9_\[\]:().]+)$")


def inline_ok(op, int_op):
    op = op.strip()
    if re.fullmatch(r"[vs]\d+", op) or op in ("vcc_lo", "vcc_hi", "m0", "exec_lo", "exec_hi"):
        return True
    if re.fullmatch(r"-?\d+", op):                     # integer inline constants -16..64
        return -16 <= int(op) <= 64
    if op in INLINE_F:
        return not int_op or op == "0"
    return False                                       # hex / other literals: keep the VOP2 form


def convert(line, stats):
    m = re.match(r"^(\s*)(v_[a-z0-9_]+)_e32(\s+)(.*?)(\s*(;.*)?)$", line)
    if not m:
        return line
    ind, name, sp, ops, tail = m.group(1), m.group(2), m.group(3), m.group(4), m.group(5)
    parts = [p.strip() for p in ops.split(",")]
    if name in PLAIN and len(parts) == 3:
        if all(inline_ok(p, name.endswith(("u32", "b32"))) for p in parts[1:]):
            stats[name] = stats.get(name, 0) + 1
            return "%s%s_e64%s%s%s\n" % (ind, name, sp, ", ".join(parts), tail)
    if name == "v_fmac_f32" and len(parts) == 3:
        if all(inline_ok(p, False) for p in parts[1:]):
            stats[name] = stats.get(name, 0) + 1
            return "%sv_fma_f32%s%s, %s, %s, %s%s\n" % (ind, sp, parts[0], parts[1], parts[2], parts[0], tail)
    stats["kept " + name] = stats.get("kept " + name, 0) + 1
    return line


def main():
    src, dst = sys.argv[1], sys.argv[2]
    stats = {}
    with open(src) as f, open(dst, "w") as g:
        for line in f:
            g.write(convert(line, stats))
    conv = {k: v for k, v in stats.items() if not k.startswith("kept ")}
    print("e64.py: re-encoded %d instructions %s" % (sum(conv.values()), conv), file=sys.stderr)


if __name__ == "__main__":
    main()
