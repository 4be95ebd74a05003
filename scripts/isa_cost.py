#!/usr/bin/env python3
"""Static issue-cost tally of a kernel's basic blocks from hipcc's assembly (-S output).

    python3 scripts/isa_cost.py file.s <kernel-name-substring> [min_block_size]

Every vector instruction is put in an issue class measured by the device micro-probe
(rt_probe_valu_issue / scripts/issue_probe.hip) at 4 waves per SIMD:
  fast  2 SIMD cycles: v_add/sub/mul/fma/fmac/mac f32, v_add/sub u32, and/or/xor/bitop3, lshrrev, mov
  slow  4 SIMD cycles: everything else (packed f32, min/max/med3, compares, selects, conversions,
        three-operand integer forms, DPP, readlane, any VOP with a scalar-register source)
  trans 8 SIMD cycles: rcp, sqrt, rsq, exp, log, sin, cos
A slow instruction holds its unit for 4 cycles but the issue port only for 2, so a block's pipe
time is max(2 * all, 4 * slow + 8 * trans)."""
import re
import sys

FAST = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|madak|madmk|fmaak|fmamk)_f32|^v_(add|sub|subrev)_u32|^v_(and|or|xor|bitop3|not)_b32|^v_lshrrev_b32|^v_mov_b32_e32|^v_mov_b64")
TRANS = re.compile(r"^v_(rcp|sqrt|rsq|exp|log|sin|cos)_")


def classify(line):
    op = line.split()[0]
    if not op.startswith("v_"):
        if op.startswith("s_"):
            return "salu"
        if op.startswith("ds_"):
            return "lds"
        return "vmem"
    if TRANS.match(op):
        return "trans"
    if "dpp" in line or "sdwa" in op:
        return "slow"
    if FAST.match(op):
        # a scalar-register source costs the fast path
        ops = line.split(None, 1)[1] if " " in line else ""
        srcs = ops.split(",")[1:]
        if any(re.match(r"\s*-?\|?s\[?\d", s) or re.match(r"\s*(vcc|exec)", s) for s in srcs):
            return "slow"
        return "fast"
    return "slow"


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_size = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().endswith(":") is False and ":" in l)
    blocks, cur, label = [], [], "entry"
    for i in range(start + 1, len(lines)):
        l = lines[i].strip()
        if l.startswith("s_endpgm"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        if re.match(r"^(v_|s_|ds_|global_|buffer_|scratch_|flat_)", l):
            cur.append((i + 1, l))
    blocks.append((label, cur))
    tot = {}
    print(f"{'block':12s} {'line':>6s} {'n':>4s} {'fast':>4s} {'slow':>4s} {'trn':>3s} {'salu':>4s} {'lds':>3s} {'vmem':>4s} {'pipe cycles':>11s}")
    for label, ins in blocks:
        c = {}
        for _, l in ins:
            k = classify(l)
            c[k] = c.get(k, 0) + 1
            tot[k] = tot.get(k, 0) + 1
        if len(ins) >= min_size:
            nv = c.get("fast", 0) + c.get("slow", 0) + c.get("trans", 0)
            pipe = max(2 * nv, 4 * c.get("slow", 0) + 8 * c.get("trans", 0))
            print(f"{label:12s} {ins[0][0]:6d} {len(ins):4d} {c.get('fast',0):4d} {c.get('slow',0):4d} {c.get('trans',0):3d} "
                  f"{c.get('salu',0):4d} {c.get('lds',0):3d} {c.get('vmem',0):4d} {pipe:11d}")
    print("total", tot)


main()
