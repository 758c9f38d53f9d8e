#!/usr/bin/env python3
"""valu_roofline.py — prices the hot loop of the fused stage-A kernel instruction by instruction.

    hipcc -O3 -std=c++20 --offload-arch=gfx950 --cuda-device-only -S -o multi.s metalign_amd/csrc/mg_sketch_multi.hip
    tools/valu_roofline.py multi.s profiles/r03/valu_classes.json [--kernel KListIJLi21ELi31ELi51] [--out profiles/r03/k1_valu_roofline.json]

Inputs: the kernel's assembly (static opcode counts of the clean-tile walk's loop: the blocks that hold the hashes, one per
k, plus the loop's own head) and tools/ubench_valu.hip's table (cycles per wave-instruction per SIMD of every opcode, in the
operand forms the compiler emits; the 8-waves-per-SIMD column = the issue cost with latencies covered).
Output: per opcode count x cycles for a position where every k is complete, the same for the positions where only the
smaller k are (the ks ascend: 21 | 21,31 | 21,31,51), the position mix of a 150-base read, and the two numbers bench.py
needs: VALU instructions per wave-step and ISSUE CYCLES per VALU instruction of this mix.  bench.py then computes
    valu_frac = SQ_INSTS_VALU x cycles_per_valu_instruction / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)
from the counters of the committed PMC pass (same kernel binary)."""
import argparse
import collections
import json
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
import isa_histogram as ih  # noqa: E402

# opcode -> row of the ubench table; VOP2 ops that issue in 2 cycles do so only without an SGPR source
FAST_VOP2 = {"v_xor_b32_e32": "v_xor_b32_e32", "v_or_b32_e32": "v_or_b32_e32", "v_and_b32_e32": "v_and_b32_e32 (literal)",
             "v_add_u32_e32": "v_add_u32_e32", "v_sub_u32_e32": "v_sub_u32_e32 (vgpr src)", "v_lshrrev_b32_e32": "v_lshrrev_b32_e32",
             "v_mov_b32_e32": "v_mov_b32_e32", "v_not_b32_e32": "v_mov_b32_e32", "v_subrev_u32_e32": "v_sub_u32_e32 (vgpr src)"}
ROW = {
    "v_mul_lo_u32": "v_mul_lo_u32 (sgpr src)", "v_mul_hi_u32": "v_mul_hi_u32 (sgpr src)", "v_mad_u64_u32": "v_mad_u64_u32 (v, s, v[2])",
    "v_add3_u32": "v_add3_u32", "v_alignbyte_b32": "v_alignbyte_b32", "v_alignbit_b32": "v_alignbit_b32", "v_perm_b32": "v_perm_b32",
    "v_bfi_b32": "v_bfi_b32", "v_lshl_or_b32": "v_lshl_or_b32", "v_lshl_add_u32": "v_lshl_add_u32", "v_bfe_u32": "v_bfe_u32",
    "v_and_or_b32": "v_and_or_b32", "v_mul_u32_u24_e32": "v_mul_u32_u24_e32", "v_mad_u32_u24": "v_mad_u32_u24",
    "v_cndmask_b32_e32": "v_cndmask_b32_e64 (vcc)", "v_cndmask_b32_e64": "v_cndmask_b32_e64 (sgpr pair)",
    "v_lshlrev_b64": "v_lshlrev_b64", "v_lshrrev_b64": "v_lshrrev_b64", "v_lshl_add_u64": "v_lshl_add_u64", "v_mov_b64_e32": "v_mov_b64_e32",
    "v_lshlrev_b32_e32": "v_lshlrev_b32_e32", "v_bcnt_u32_b32": "v_bcnt_u32_b32", "v_bitop3_b32": "v_bitop3_b32",
    "v_add_co_u32_e32": "v_add_co_u32 + v_addc_co_u32", "v_addc_co_u32_e32": "v_add_co_u32 + v_addc_co_u32",
}
SGPR_FORM = {"v_xor_b32_e32": "v_xor_b32_e32 (sgpr src)", "v_add_u32_e32": "v_add_u32_e32"}  # measured with an SGPR source


def cycles_table(path, waves):
    d = json.load(open(path))
    return {r["op"]: r["cycles"] for r in d["rows"] if r["waves_per_simd"] == waves}


def price(line_op, operands, cyc):
    """(cycles, how) of one VALU instruction."""
    op = line_op
    if op in FAST_VOP2:
        sg = bool(re.search(r"(?<![a-z_0-9])(s\d+|s\[\d+:\d+\]|vcc|exec|m0)(?![a-z_0-9])", operands.split(",", 1)[1] if "," in operands else ""))
        if sg:
            return cyc.get("v_xor_b32_e32 (sgpr src)", 4.2), "VOP2 with an SGPR source"
        base = {"v_xor_b32_e32": "v_xor_b32_e32 (vgpr src)", "v_add_u32_e32": "v_add_u32_e32 (vgpr src)"}.get(op, FAST_VOP2[op])
        return cyc[base], base
    if op.startswith("v_cmp"):
        return cyc["v_cmp_ge_u64_e64" if "64" in op.split("_")[-2] + op.split("_")[-1] else "v_cmp_lt_u32_e32"], "compare"
    if op in ROW:
        return cyc[ROW[op]], ROW[op]
    if op.endswith("_sdwa") or op.endswith("_dpp"):
        return cyc["v_add3_u32"], "SDWA / DPP form: priced as a VOP3"
    return cyc["v_add3_u32"], "not measured one by one: priced as a VOP3"


INSTR = re.compile(r"^\t(v_[a-z_0-9]+)\s+(.*?)\s*(;.*)?$")


def block_lines(path, needle, labels):
    """{label: [(opcode, operands)]} of the VALU instructions of the named blocks."""
    out, cur, inside = {l: [] for l in labels}, None, False
    with open(path) as fh:
        for line in fh:
            if not inside:
                inside = line.startswith("_Z") and needle in line.split(":")[0]
                cur = "entry"
                continue
            if line.startswith(".Lfunc_end"):
                break
            m = ih.LABEL.match(line)
            if m:
                cur = m.group(1)
                continue
            m = INSTR.match(line)
            if m and cur in out:
                out[cur].append((m.group(1), m.group(2)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("classes")
    ap.add_argument("--kernel", default="20k_sketch_reads_multiINS_5KListIJLi21ELi31ELi51EEEELi0E")  # (not ..._resident)
    ap.add_argument("--waves", type=int, default=8, help="column of the ubench table (8: issue cost with latencies covered)")
    ap.add_argument("--read_len", type=int, default=150)
    ap.add_argument("--ks", default="21,31,51")
    ap.add_argument("--out", default="")
    ap.add_argument("--positions_per_iteration", type=int, default=1,
                    help="read positions one iteration of the walk loop covers (the one-k kernels, mg_sketch_kernel.h walk_reads: 2)")
    a = ap.parse_args()
    ks = [int(x) for x in a.ks.split(",")]
    cyc = cycles_table(a.classes, a.waves)
    blocks, order, succ = ih.parse_kernel(a.asm, a.kernel)
    idx = {b: i for i, b in enumerate(order)}
    # the walk loops: back edges whose body holds multiplies; per loop, the blocks with multiplies are the hashes (one per
    # k, in ascending k) — the clean equal-length walk (MODE 1) is the loop without a per-lane length compare in its hashes
    # the walk loops: back edges whose body holds multiplies.  A loop over the positions where EVERY k is complete holds all
    # the multiplies of a position (whatever the number of blocks: the shipped kernel hashes two k per block there); a loop
    # over the positions where the largest k is not complete yet holds fewer, one block per k.  There is one of each per
    # tile walk; the clean equal-length walk (MODE 1) is the one with the fewest 32-bit compares (no `pos < len`, no run
    # counter).
    def nmul(bs):
        return sum(1 for b in bs for o in blocks[b] if ih.classify(o) in ("valu_mul32", "valu_mad64"))
    loops = []
    for head, tail, body in ih.loops(blocks, order, succ):
        hot = [b for b in body if any(ih.classify(o) in ("valu_mul32", "valu_mad64") for o in blocks[b])]
        if hot and len(body) < 400 and nmul(hot) > 30:
            loops.append((head, tail, body, hot))
    if not loops:
        sys.exit("no walk loop found")
    top = max(nmul(c[3]) for c in loops)
    cands = [c for c in loops if nmul(c[3]) == top]
    paired = [c for c in loops if nmul(c[3]) < top and len(c[3]) == len(ks) - 1]
    lines_all = {}
    for head, tail, body, hot in cands:
        lines_all[head] = block_lines(a.asm, a.kernel, [head] + hot)
    # MODE 1 = the candidate with the fewest v_cmp on 32 bits in its hash blocks (no `pos < len`)
    def ncmp32(c):  # (the clean walk has neither the per-lane length compare nor the run-counter tests)
        return sum(1 for b in c[3] for op, _ in lines_all[c[0]][b] if op.startswith("v_cmp") and ("32" in op))
    head, tail, body, hot = min(cands, key=ncmp32)
    lines = lines_all[head]
    per_block = {}
    for b in [head] + hot:
        rows = collections.OrderedDict()
        for op, operands in lines[b]:
            c, how = price(op, operands, cyc)
            key = (op, how)
            r = rows.setdefault(key, [0, c])
            r[0] += 1
        per_block[b] = rows
    # a position where k number i is the largest complete one executes the head + the hash blocks 0..i
    L = a.read_len
    npos = [max(0, min(L, (ks[i + 1] - 1) if i + 1 < len(ks) else L) - (ks[i] - 1)) for i in range(len(ks))]
    warm = min(L, ks[0] - 1)
    table, tot_instr, tot_cyc = [], 0.0, 0.0
    kinds = []
    # the positions where the largest k is not complete yet run their own loop (mg_sketch_multi.hip): the smaller k's
    # hashes are priced from THAT loop's blocks when the binary has it
    part = None
    if paired:
        def ncmp32p(c):
            ls = block_lines(a.asm, a.kernel, [c[0]] + c[3])
            return sum(1 for b in c[3] for op, _ in ls[b] if op.startswith("v_cmp") and ("32" in op))
        ph, pt, pb, phot = min(paired, key=ncmp32p)
        pl = block_lines(a.asm, a.kernel, [ph] + phot)
        part = {}
        for b in [ph] + phot:
            rows = collections.OrderedDict()
            for op, operands in pl[b]:
                c, how = price(op, operands, cyc)
                r = rows.setdefault((op, how), [0, c])
                r[0] += 1
            part[b] = rows
    for i in range(len(ks)):
        n_i = c_i = 0.0
        if part is not None and i < len(ks) - 1:
            for b in [ph] + phot[: i + 1]:
                for (op, how), (n, c) in part[b].items():
                    n_i += n
                    c_i += n * c
            kinds.append({"complete_k": ks[: i + 1], "positions_per_read": npos[i], "valu_instructions": n_i, "valu_cycles": c_i,
                          "loop": {"head": ph, "hash_blocks": phot[: i + 1]}})
            tot_instr += npos[i] * n_i
            tot_cyc += npos[i] * c_i
            continue
        for b in [head] + (hot if i == len(ks) - 1 else hot[: i + 1]):
            for (op, how), (n, c) in per_block[b].items():
                n_i += n
                c_i += n * c
        n_i /= a.positions_per_iteration
        c_i /= a.positions_per_iteration
        kinds.append({"complete_k": ks[: i + 1], "positions_per_read": npos[i], "valu_instructions": n_i, "valu_cycles": c_i})
        tot_instr += npos[i] * n_i
        tot_cyc += npos[i] * c_i
    # the warm-up positions (no k complete): the loop head only, approximately (a separate, smaller loop in the binary)
    n_w = sum(n for (op, how), (n, c) in per_block[head].items())
    c_w = sum(n * c for (op, how), (n, c) in per_block[head].items())
    tot_instr += warm * n_w
    tot_cyc += warm * c_w
    full = collections.OrderedDict()
    for b in [head] + hot:
        for (op, how), (n, c) in per_block[b].items():
            r = full.setdefault(op + " | " + how, [0, c])
            r[0] += n
    for k, (n, c) in sorted(full.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        table.append({"instruction": k, "count": n, "cycles_each": round(c, 2), "cycles": round(n * c, 1)})
    res = {"kernel": a.kernel, "loop": {"head": head, "hash_blocks": hot}, "ks": ks, "read_len": L,
           "positions_per_loop_iteration": a.positions_per_iteration,
           "ubench_column": "%d waves per SIMD" % a.waves,
           "all_k_complete_position": table, "position_kinds": kinds, "warm_up_positions_per_read": warm,
           "valu_instructions_per_wave_step": tot_instr / L, "valu_cycles_per_wave_step": tot_cyc / L,
           "cycles_per_valu_instruction": tot_cyc / tot_instr,
           "note": "static counts of the clean-tile walk (the hit path's nine instructions per k are inside the hash blocks and "
                   "counted as always executed; a flush is not in the loop's steady state); cycles from tools/ubench_valu.hip"}
    print("loop %s, hash blocks %s" % (head, hot))
    print("%-62s %6s %8s %9s" % ("instruction | priced as", "count", "cyc each", "cycles"))
    for r in table:
        print("%-62s %6d %8.2f %9.1f" % (r["instruction"][:62], r["count"], r["cycles_each"], r["cycles"]))
    for kd in kinds:
        print("positions with k in %s complete: %d per read, %.0f VALU instructions, %.0f cycles" % (kd["complete_k"], kd["positions_per_read"], kd["valu_instructions"], kd["valu_cycles"]))
    print("per wave-step (read of %d): %.1f VALU instructions, %.1f issue cycles -> %.3f cycles per VALU instruction"
          % (L, res["valu_instructions_per_wave_step"], res["valu_cycles_per_wave_step"], res["cycles_per_valu_instruction"]))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
