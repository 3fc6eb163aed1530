#!/usr/bin/env python3
"""Developer tool: the instruction mix of the two-lane path kernel's vote loop by STEP KIND, from the ISA.

    tools/isa_mix.py [node_steps leaf_steps done_steps]      (wave-level steps per bounce, default: tools/travsim.cpp on workload C2)

Compiles csrc/trace_kernels.hip with -DRVB_ISA_MARKS=1 (comment lines at the borders of the vote / node / leaf / shading blocks of
traverse_pairs_vote; the marked build is never shipped: the markers are volatile asm and pin the block order), classifies every
instruction of path_pair_kernel between the markers and weights the three step kinds with the wave-level step counts of one bounce.
Answers what the PMC class counters cannot: what the "other" VALU instructions (neither f32 add / mul / fma nor integer) ARE."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "parallel-reverb-raytracer_amd")

CLASSES = [
    ("f32 add / mul / fma (incl. fma_mix from binary16 planes)", r"^v_(fma_f32|fma_mix_f32|mul_f32|add_f32|sub_f32|subrev_f32|fmac_f32|mul_legacy_f32|mad_f32)"),
    ("f32 min / max (slab test, culling)", r"^v_(min|max|min3|max3|med3)_f32"),
    ("f32 reciprocal / sqrt", r"^v_(rcp|rsq|sqrt|rcp_iflag)_f32"),
    ("compare", r"^v_cmp"),
    ("select (v_cndmask)", r"^v_cndmask"),
    ("DPP move / DPP min (lane exchange inside a ray's lanes)", r"_dpp"),
    ("byte permute (near / far plane selection)", r"^v_perm_b32"),
    ("bit logic / shifts / popcount / masks", r"^v_(and|or|xor|not|lshl|lshr|ashr|bfe|bfm|bfi|bcnt|and_or|or3|lshl_or|lshl_add|bitop3|alignbit)"),
    ("integer add / mul / 64-bit multiply-add", r"^v_(add_u32|sub_u32|subrev_u32|add3_u32|add_co|addc_co|sub_co|mul_lo|mul_u32_u24|mul_hi|mad_u64_u32|mad_u32_u24|add_lshl_u32|min_u32|max_u32|min_i32|max_i32|mad_i32_i24)"),
    ("move / readlane", r"^v_(mov|readfirstlane|readlane|writelane|accvgpr|swap)"),
    ("convert", r"^v_cvt"),
    ("binary64 (exp of the air attenuation is in the shadow kernel; here: none expected)", r"_f64"),
    ("LDS (stack, surface table)", r"^ds_"),
    ("vector memory", r"^(global|buffer|flat|scratch)_"),
    ("scalar (vote, masks, branches, waits)", r"^s_"),
]


def classify(op):
    for name, pattern in CLASSES:
        if re.search(pattern, op):
            return name
    return "unclassified: " + op


# Issue cost of a vector instruction relative to v_fma_f32 with register operands, 8 waves per SIMD (tools/inst_probe.hip,
# profiles/r04b_inst_probe.log): the binary32 add / mul / fma, the move, the two-operand integer add / sub, and / or / xor, the right
# shift and v_bitop3_b32 issue at the fast rate, every other form (comparisons, selects, min / max, DPP, byte permute, fma_mix, the
# three-operand integer forms, anything that reads an SGPR in a three-operand form) takes 1.6-1.8 times as long.
FAST = re.compile(r"^v_(fma_f32|fmac_f32|add_f32|sub_f32|subrev_f32|mul_f32|mul_legacy_f32|mov_b32|bitop3_b32|and_b32|or_b32|xor_b32|lshrrev_b32|"
                  r"add_u32|sub_u32|subrev_u32)(_e32|_e64)?$")


def issue_cost(line):
    op = line.split()[0]
    if not op.startswith("v_"):
        return 0.0
    if re.match(r"^v_(rcp|rsq|sqrt)_f32", op):
        return 3.1
    if re.match(r"^v_rcp_f64", op):
        return 6.2
    if FAST.match(op) and "dpp" not in line and "sdwa" not in line:
        operands = line.split(None, 1)[1] if len(line.split(None, 1)) > 1 else ""
        if op.startswith("v_fma_f32") and re.search(r"(^|[ ,\-|])s(\d|\[)", operands):
            return 1.75                  # a scalar operand in the three-operand form
        return 1.0
    return 1.7


def main():
    steps = [float(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else [23.73, 6.30, 3.18]      # travforms (TRAVFORMS_CYCLE=34,25), C2, 32 rays per wave; the majority vote of rounds 1-3: 28.66 5.14 2.57
    asm = "/tmp/isa_mix.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "-fhip-fp32-correctly-rounded-divide-sqrt", "-munsafe-fp-atomics", "-fno-slp-vectorize", "--cuda-device-only",
                           "-DRVB_ISA_MARKS=1", *os.environ.get("ISA_MIX_FLAGS", "").split(), "-S", "-o", asm, os.path.join(PKG, "csrc", "trace_kernels.hip")], cwd=PKG)
    text = open(asm).read()
    start = text.index("_ZN12_GLOBAL__N_122path_pair_group_kernelILb1EEEvNS_10TraceGroupE:")
    body = text[start:text.index("s_endpgm", start)]
    blocks, current = collections.OrderedDict(), None
    cost = collections.Counter()
    for line in body.splitlines():
        line = line.strip()
        m = re.match(r"; RVB_MARK (\w+)", line)
        if m:
            current = m.group(1)
            blocks.setdefault(current, collections.Counter())
            continue
        if current in (None, "loop_end") or not line or line.startswith((";", ".", "//")) or line.endswith(":"):
            continue
        op = line.split()[0]
        if op in ("s_nop",) or op.startswith(";;#"):
            continue
        blocks[current][classify(op)] += 1
        cost[current] += issue_cost(line)
    weights = {"vote": steps[0], "node": steps[0], "leaf": steps[1], "done": steps[2]}      # one loop iteration per node step
    print("static instruction counts per block of traverse_pairs_vote (path_pair_kernel<true>), and per bounce of one wave (32 rays)")
    print("weights = wave-level executions per bounce: %s" % weights)
    total = collections.Counter()
    for name, counter in blocks.items():
        n = sum(counter.values())
        valu = sum(v for k, v in counter.items() if not k.startswith(("LDS", "vector memory", "scalar")))
        print("\n[%s] %d instructions, %d of them VALU = %.1f issue units (v_fma_f32 = 1), x %.2f per bounce" % (name, n, valu, cost[name], weights.get(name, 0.0)))
        for k, v in counter.most_common():
            print("   %4d  %s" % (v, k))
            total[k] += v * weights.get(name, 0.0)
    valu_total = sum(v for k, v in total.items() if not k.startswith(("LDS", "vector memory", "scalar")))
    print("\nper bounce of one wave, all blocks weighted: %.0f instructions, %.0f VALU, %.0f issue units" % (sum(total.values()), valu_total, sum(cost[k] * weights.get(k, 0.0) for k in cost)))
    for k, v in total.most_common():
        share = "" if k.startswith(("LDS", "vector memory", "scalar")) else "  (%.1f %% of VALU)" % (100.0 * v / valu_total)
        print("   %7.0f  %s%s" % (v, k, share))


if __name__ == "__main__":
    main()
