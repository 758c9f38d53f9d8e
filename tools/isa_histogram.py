#!/usr/bin/env python3
"""isa_histogram.py — basic blocks of a gfx950 kernel's assembly (hipcc --cuda-device-only -S), their instruction mix by
issue class, and the loops among them.

    hipcc -O3 -std=c++20 --offload-arch=gfx950 --cuda-device-only -S -o multi.s metalign_amd/csrc/mg_sketch_multi.hip
    tools/isa_histogram.py multi.s 'KListIJLi21ELi31ELi51' [--blocks] [--loop LBB0_123] [--json out.json]

Issue classes (what tools/ubench_valu.hip prices one by one on the GPU box; profiles/r03/valu_classes.json):
    valu_1op     32-bit VALU, VOP1/VOP2/VOPC encodings and plain VOP3 forms of them (v_xor, v_and, v_or, v_add_u32, v_lshl..)
    valu_3src    three-source 32-bit VOP3 (v_alignbit, v_alignbyte, v_perm, v_bfi, v_add3, v_lshl_or, v_and_or, v_xad, v_bfe)
    valu_carry   v_add_co / v_addc_co / v_sub_co / v_subb_co (VOP3b / VOP2 with VCC)
    valu_mul32   v_mul_lo_u32 / v_mul_hi_u32
    valu_mad64   v_mad_u64_u32
    valu_64      64-bit shifts / adds (v_lshlrev_b64, v_lshrrev_b64, v_lshl_add_u64, v_cmp_*_u64)
    valu_cmp     v_cmp* on 32 bits, v_cndmask
    valu_xlane   v_readlane / v_readfirstlane / v_writelane / DPP / v_mbcnt / v_permlane
    lds, vmem, salu, smem, branch, wait, other
"""
import argparse
import collections
import json
import re
import sys

THREE_SRC = ("v_alignbit_b32", "v_alignbyte_b32", "v_perm_b32", "v_bfi_b32", "v_add3_u32", "v_lshl_or_b32", "v_and_or_b32",
             "v_or3_b32", "v_xad_u32", "v_bfe_u32", "v_bfe_i32", "v_lshl_add_u32", "v_add_lshl_u32", "v_xor3_b32", "v_mad_u32_u24",
             "v_mad_i32_i24", "v_med3", "v_min3", "v_max3")


def classify(op):
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
        return "valu_mad64"
    if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_lo_i32", "v_mul_hi_i32")):
        return "valu_mul32"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_mbcnt", "v_permlane", "v_bpermute")) or "_dpp" in op:
        return "valu_xlane"
    if op.startswith(("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_lshl_add_u64")) or re.match(r"v_cmpx?_\w+_[ui]64", op):
        return "valu_64"
    if op.startswith(("v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_subrev_co", "v_subbrev_co")):
        return "valu_carry"
    if op.startswith(THREE_SRC):
        return "valu_3src"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "valu_cmp"
    if op.startswith("v_"):
        return "valu_1op"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_barrier")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


INSTR = re.compile(r"^\t([a-z_0-9]+)(\s|$)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")


def parse_kernel(path, needle):
    """-> ordered {label: [opcodes]}, successors {label: set(labels)}"""
    blocks, order, cur, inside = {}, [], None, False
    succ = collections.defaultdict(set)
    with open(path) as fh:
        for line in fh:
            if not inside:
                if line.startswith("_Z") and needle in line.split(":")[0]:
                    inside, cur = True, "entry"
                    blocks[cur] = []
                    order.append(cur)
                continue
            if line.startswith(".Lfunc_end"):
                break
            m = LABEL.match(line)
            if m:
                nxt = m.group(1)
                last = blocks[cur][-1] if blocks[cur] else ""
                if not last.startswith(("s_branch", "s_endpgm", "s_setpc")):
                    succ[cur].add(nxt)  # fall through
                cur = nxt
                blocks[cur] = []
                order.append(cur)
                continue
            m = INSTR.match(line)
            if not m or line.startswith("\t."):
                continue
            op = m.group(1)
            blocks[cur].append(op)
            if op.startswith(("s_cbranch", "s_branch")):
                t = line.split()[-1]
                if t.startswith(".LBB"):
                    succ[cur].add(t)
    return blocks, order, succ


def histogram(ops):
    h = collections.Counter(classify(o) for o in ops)
    return dict(sorted(h.items()))


def loops(blocks, order, succ):
    """Natural loops by back edges in layout order: (head, tail) with tail at or after head -> the blocks in between."""
    idx = {b: i for i, b in enumerate(order)}
    out = []
    for b in order:
        for t in succ[b]:
            if t in idx and idx[t] <= idx[b]:
                out.append((t, b, order[idx[t]: idx[b] + 1]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("needle", help="substring of the mangled kernel name")
    ap.add_argument("--blocks", action="store_true", help="list every basic block")
    ap.add_argument("--loop", default="", help="head label of the loop to report in detail (default: the loop with the most multiplies among innermost loops)")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    blocks, order, succ = parse_kernel(a.asm, a.needle)
    if not blocks:
        sys.exit("kernel not found")
    if a.blocks:
        for b in order:
            h = histogram(blocks[b])
            print(b, len(blocks[b]), h, "->", sorted(succ[b]))
    ls = loops(blocks, order, succ)
    rows = []
    for head, tail, body in ls:
        ops = [o for b in body for o in blocks[b]]
        nmul = sum(1 for o in ops if classify(o) in ("valu_mul32", "valu_mad64"))
        rows.append(dict(head=head, tail=tail, nblocks=len(body), ninstr=len(ops), nmul=nmul, hist=histogram(ops),
                         opcodes=dict(collections.Counter(ops).most_common())))
    rows.sort(key=lambda r: -r["nmul"])
    print("loops (head, tail, blocks, instructions, multiplies):")
    for r in rows[:24]:
        print("  %-12s %-12s %4d %6d %5d" % (r["head"], r["tail"], r["nblocks"], r["ninstr"], r["nmul"]))
    pick = next((r for r in rows if r["head"] == a.loop), None) if a.loop else None
    if pick:
        print(json.dumps({k: pick[k] for k in ("head", "tail", "ninstr", "hist")}, indent=1))
        print(json.dumps(pick["opcodes"], indent=1))
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
