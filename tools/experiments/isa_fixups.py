#!/usr/bin/env python3
"""isa_fixups.py — rewrites of the gfx950 assembly hipcc emits for this library's kernels, applied between
`hipcc --cuda-device-only -S` and the assembler (see the Makefile).  Usage: isa_fixups.py in.s out.s

1. v_cndmask_b32_e32 (VOP2, mask = implicit VCC)  ->  v_cndmask_b32_e64 (VOP3, mask operand vcc).
   Measured on MI355X (tools/ubench_valu.hip, profiles/r03/valu_classes.json): the VOP2 encoding of the select issues at
   22.5 cycles per wave-instruction per SIMD whatever the occupancy (1..8 waves per SIMD), dependent or independent,
   VCC fresh or written long before — the VOP3 encoding of the SAME operation, mask in VCC or in any SGPR pair, at 4.2
   like every other VOP3.  LLVM's SIShrinkInstructions turns every select whose mask can live in VCC into the VOP2
   form (4 bytes instead of 8) and has no switch; stage C (k_profile_pass) had 200 of them among 2780 vector
   instructions, k_contain_pairs 38 of 469.  The rewrite keeps operands and order, so the wait states the compiler put
   between a VALU write of VCC and the select (s_nop 1) still stand; only selects whose first source is a LITERAL stay
   as they are (VOP3 on gfx9 encodes no literal).
The script prints how many instructions it rewrote; the kernels' results are covered by the parity tests as before."""
import re
import sys

CND = re.compile(r"^(\s*)v_cndmask_b32_e32(\s+)(v\d+),\s*([^,]+),\s*(v\d+),\s*vcc(\s*(?:;.*)?)$")
FLOAT_INLINE = {"0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0"}


def vop3_can_encode(src0):
    src0 = src0.strip()
    if re.fullmatch(r"[vs]\d+", src0) or src0 in ("vcc_lo", "vcc_hi", "m0", "exec_lo", "exec_hi") or src0 in FLOAT_INLINE:
        return True
    try:
        v = int(src0, 0)
    except ValueError:
        return False
    return -16 <= v <= 64


def main(src, dst):
    n = kept = 0
    out = []
    with open(src) as fh:
        for line in fh:
            m = CND.match(line.rstrip("\n"))
            if m:
                if vop3_can_encode(m.group(4)):
                    line = "%sv_cndmask_b32_e64%s%s, %s, %s, vcc%s\n" % (m.group(1), m.group(2), m.group(3), m.group(4).strip(), m.group(5), m.group(6))
                    n += 1
                else:
                    kept += 1
            out.append(line)
    with open(dst, "w") as fh:
        fh.writelines(out)
    print("isa_fixups: %s: %d v_cndmask_b32_e32 -> e64, %d left (literal source)" % (src, n, kept))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
