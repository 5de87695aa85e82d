#!/usr/bin/env python3
"""Re-encode `v_cndmask_b32_e32 vD, src0, vS1, vcc` (VOP2, implicit VCC) as `v_cndmask_b32_e64 vD, src0, vS1, vcc` (VOP3) in gfx950 assembly.

Why: measured on MI355X (profiles/tools/ubench/valu_sel.hip, profiles/round2/ubench_select.txt) the VOP2 encoding of v_cndmask_b32 issues at
9.8 ns per wave-instruction per SIMD, the VOP3 encoding of the SAME operation (mask in VCC or in any SGPR pair) at 1.87 ns — every FP64
select of the hydro / radiation kernels is two of them, and hipcc shrinks them to VOP2 whenever the mask lands in VCC.  The two encodings
are the same instruction (same operands, same result, same hazards: the s_nop the compiler placed after the VCC writer stays in place), only
4 bytes longer; branch offsets are recomputed by the assembler.

Not convertible (left as they are): src0 an SGPR or a 32-bit literal — VOP3 on gfx9 allows one constant-bus operand and VCC already is one,
and VOP3 takes no literals; SDWA / DPP forms.
usage: vop3_cndmask.py in.s out.s   (prints the counts)"""
import re
import sys

INLINE = re.compile(r"^(-?\d+|-?0\.5|-?1\.0|-?2\.0|-?4\.0|0\.15915494)$")
PAT = re.compile(r"^(\s*)v_cndmask_b32_e32(\s+)(v\d+),\s*([^,]+),\s*(v\d+),\s*vcc\s*(;.*)?$")


def convertible(src0: str) -> bool:
    src0 = src0.strip()
    if re.fullmatch(r"v\d+", src0):
        return True
    if INLINE.match(src0):
        try:
            v = float(src0)
        except ValueError:
            return False
        if src0.lstrip("-").isdigit():
            return -16 <= int(src0) <= 64
        return True
    return False


def main():
    src, dst = sys.argv[1], sys.argv[2]
    done = kept = 0
    out = []
    for line in open(src):
        m = PAT.match(line.rstrip("\n"))
        if m and convertible(m.group(4)):
            out.append(f"{m.group(1)}v_cndmask_b32_e64{m.group(2)}{m.group(3)}, {m.group(4).strip()}, {m.group(5)}, vcc\n")
            done += 1
        else:
            if "v_cndmask_b32_e32" in line:
                kept += 1
            out.append(line)
    open(dst, "w").writelines(out)
    print(f"vop3_cndmask: {src}: {done} re-encoded, {kept} left as VOP2")


if __name__ == "__main__":
    main()
