#!/usr/bin/env python3
"""Instruction ledger of one kernel's hot loop from a `hipcc -S --cuda-device-only` listing: every basic block's
instructions by issue class, and the weighted sum over the blocks named on the command line -- priced with the issue
costs measured on MI355X (tools/valu_bench*.hip, cycles per wave-instruction per SIMD at >= 2 waves per SIMD).

usage: asm_ledger.py file.s <kernel-name-substring> [LABEL=weight | FIRST..LAST=weight ...]
       (no labels: list every block >= 20 instructions; FIRST..LAST names every block of that textual range, later
        arguments override earlier ones)

The priced sum is a LOWER bound on the SIMD cycles the loop needs (every instruction issued back to back at its own
best rate), so `priced cycles / measured cycles` cannot exceed 1 -- unlike 4 SQ_ACTIVE_INST_VALU / SIMD cycles, which
counts overlapping execution of two waves twice."""
import re
import sys
from collections import Counter, OrderedDict

# cycles per wave-instruction per SIMD at EXACTLY two resident waves (256-register kernels), long runs, quoted at 2.4 GHz
# (i.e. times: 1 cycle = 0.4167 ns) -- tools/pk_vs_fma_2waves.hip, round 4.  (tools/valu_bench.hip's "waves/SIMD" rows let
# the dispatcher pack small kernels unevenly; its figures were 4.4 / 8.2 / 4.2 / 5.4 / 4.2 / 2.6.)
COST = OrderedDict([("pk_f32", 4.43), ("trans", 8.17), ("dpp", 4.41), ("mad_u64", 4.42), ("half_rate", 4.19), ("fma3", 3.02),
                    ("mov", 2.48), ("valu_other", 2.48)])
TRANS = ("v_exp_", "v_log_", "v_sqrt_", "v_rsq_", "v_rcp_", "v_sin_", "v_cos_")
HALF = ("v_lshl", "v_lshr", "v_ashr", "v_alignbit", "v_cvt_", "v_min_", "v_max_", "v_cmp", "v_cndmask", "v_bfi", "v_bfe",
        "v_readlane", "v_writelane", "v_readfirstlane", "v_mul_lo_u32", "v_mul_hi_u32")


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        if "_dpp" in op or " dpp" in ins or "quad_perm" in ins or "row_" in ins:
            return "dpp"
        if op.startswith("v_pk_"):
            return "pk_f32"
        if op.startswith(TRANS):
            return "trans"
        if op.startswith("v_mad_u64") or op.startswith("v_mad_i64"):
            return "mad_u64"
        if op.startswith("v_mov_") or op.startswith("v_accvgpr"):
            return "mov"
        if op.startswith(HALF):
            return "half_rate"
        if op.startswith("v_mfma"):
            return "mfma"
        if op.startswith(("v_fma_f32", "v_fmamk_f32", "v_fmaak_f32")):
            return "fma3"          # three source operands: 3.0 cycles against 2.5 for the two-source forms
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op in ("s_waitcnt", "s_nop", "s_barrier", "s_sleep"):
        return "wait_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2]
    wargs = [a.split("=") for a in sys.argv[3:]]
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, label, cur = OrderedDict(), "entry", []
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks[label] = cur
            label, cur = m.group(1), []
        elif t and not t.startswith((".", ";", "//")):
            cur.append(t)
    blocks[label] = cur
    weights = OrderedDict()
    order = list(blocks)
    for k, v in wargs:
        if ".." in k:
            a, b = k.split("..")
            for lab in order[order.index(a):order.index(b) + 1]:
                weights[lab] = float(v)
        else:
            weights[k] = float(v)
    classes = list(COST) + ["mfma", "salu", "lds", "vmem", "wait_nop", "other"]
    print("kernel: %s" % lines[start].split(":")[0])
    print("%-12s %6s  " % ("block", "instr") + " ".join("%10s" % c for c in classes))
    tot = Counter()
    for lab, ins in blocks.items():
        if (weights and (lab not in weights or weights[lab] == 0.0 or not ins)) or (not weights and len(ins) < 20):
            continue
        c = Counter(classify(i) for i in ins)
        w = weights.get(lab, 1.0)
        print("%-12s %6d  " % (lab + ("" if not weights else " x%g" % w), len(ins)) + " ".join("%10d" % c[k] for k in classes))
        for k in classes:
            tot[k] += w * c[k]
    if weights:
        print("%-12s %6.0f  " % ("weighted", sum(tot.values())) + " ".join("%10.1f" % tot[k] for k in classes))
        valu = sum(tot[k] for k in COST)
        priced = sum(tot[k] * COST[k] for k in COST)
        print("VALU instructions per loop iteration: %.1f; priced at %s cycles per wave-instruction: %.0f SIMD cycles per wave"
              % (valu, dict(COST), priced))


if __name__ == "__main__":
    main()
