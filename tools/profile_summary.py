#!/usr/bin/env python3
"""Summarises the raw rocprofv3 output of tools/profile.sh <tag> (gpurun_out/<tag>_*) into
gpurun_out/<tag>_kernel_stats.csv (our kernels + the sort) and gpurun_out/<tag>_pmc.json (per-launch averages of every
counter per kernel, HBM bytes with the gfx950 FETCH_SIZE correction, VALU issue share)."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
SHORT = ["path_kernel", "path_lane_group_kernel", "path_pair_group_kernel", "path_pair_kernel", "image_plan_kernel", "image_check_kernel", "shadow_kernel", "shadow_pair_kernel",
         "bin_keys_hrtf_kernel", "ordered_sum_hrtf_kernel", "histogram_fast_kernel", "histogram_transpose_kernel", "attenuate_kernel",
         "time_range_kernel", "bin_keys_kernel", "ordered_sum_kernel", "radix_sort_onesweep_iteration", "radix_sort_onesweep_global_offsets"]

# VALU issue model: dynamic instruction mix (per-class PMC counters) x measured issue cost per class (tools/inst_probe.hip at
# 8 waves per SIMD, ns per wave-instruction per SIMD, parsed from profiles/<tag>_inst_probe.log by tools/inst_costs.py)
CLASSES = {
    "f32_add_mul_fma": (["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"], ["add_f32", "mul_f32", "v_fma_f32"]),
    "f32_transcendental": (["SQ_INSTS_VALU_TRANS_F32"], ["rcp", "sqrt"]),
    "f64_add_mul_fma": (["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"], ["add_f64", "mul_f64", "fma_f64"]),
    "f64_transcendental": (["SQ_INSTS_VALU_TRANS_F64"], ["rcp_f64"]),
    "convert": (["SQ_INSTS_VALU_CVT"], ["cvt_f16", "cvt_f32_f64", "cvt_f64_f32", "cvt_i32_f64"]),
    "int32": (["SQ_INSTS_VALU_INT32"], ["and_or", "lshl_add", "mul_lo", "mul_u24", "bcnt"]),
    "int64": (["SQ_INSTS_VALU_INT64"], ["cmp_u64"]),
    # everything the class counters do not name: min / max / compare / select / DPP and plain moves
    "other": ([], ["min", "max3", "cmp", "cndmask_sgpr", "mov_dpp", "min_dpp", "perm", "mov"]),
}
SIMDS = 1024


def short(name):
    for s in SHORT:
        if s in name:
            return s
    return None


# kernel stats: keep the full csv, print ours
for suffix in ("", "_default"):
    stats = glob.glob(os.path.join(out, tag + "_stats" + suffix, "**", "*kernel_stats.csv"), recursive=True)
    if not stats:
        continue
    print("--- kernel stats%s" % suffix)
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, tag + "_kernel_stats%s.csv" % suffix), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                s = short(r["Name"])
                w.writerow([s or r["Name"][:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
                if s:
                    print("%-36s calls %3s avg %10.1f us" % (s, r["Calls"], float(r["AverageNs"]) / 1e3))

pmc = {}
for d in sorted(glob.glob(os.path.join(out, tag + "_pmc_*"))):
    if not os.path.isdir(d):
        continue
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(path)):
            s = short(r["Kernel_Name"])
            if not s:
                continue
            key = (s, r["Counter_Name"])
            a = acc.setdefault(key, [0.0, set()])
            a[0] += float(r["Counter_Value"])
            a[1].add(r["Dispatch_Id"])
        for (s, c), (total, disp) in acc.items():
            pmc.setdefault(s, {})[c] = total / max(1, len(disp))
            pmc[s].setdefault("_launches", len(disp))
            if c == "SQ_INSTS_VALU":
                pmc[s]["SQ_INSTS_VALU_all_launches"] = total
# FETCH_SIZE tallies a wide coalesced streaming read at HALF its bytes and a gather of 64-byte records at its TRUE bytes (tools/fetch_calibration.hip,
# profiles/r04_fetch_calibration_n1.txt: 1 GiB streamed -> 512 MiB, 512 MiB of random 64-byte records -> 510 MiB, 1 GiB of records read and rewritten in
# place -> 1 067 MiB fetched, 1 072 MiB written): the x2 of the guide applies to the STREAMING kernels only
STREAMING = {"radix_sort_onesweep_iteration", "radix_sort_onesweep_global_offsets", "bin_keys_kernel", "bin_keys_hrtf_kernel", "bin_starts_kernel", "attenuate_kernel",
             "histogram_fast_kernel", "histogram_transpose_kernel", "time_range_kernel"}
for s, c in pmc.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0        # the guide's correction for every kernel: an upper bound for gathers
        c["fetch_factor_calibrated"] = 2.0 if s in STREAMING else 1.0
        c["hbm_bytes_calibrated_per_launch"] = (c["fetch_factor_calibrated"] * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        c["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"] > 0:
        # SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units")
        c["valu_active_share_of_wave_cycles"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
        c["wait_any_share_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c else None
costs = {}
try:
    costs = json.load(open(os.path.join(out, tag + "_inst_costs.json")))["8"]
except (OSError, KeyError, ValueError):
    pass
for s_, c in pmc.items():
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        # rocprofiler's VALUUtilization: active lanes per VALU instruction / 64
        c["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
    if costs and "SQ_INSTS_VALU" in c and "SQ_INSTS_VALU_INT32" in c:
        mix, named, model_ns = {}, 0.0, 0.0
        for name, (counters, probes) in CLASSES.items():
            if not counters:
                continue
            n = sum(c.get(k, 0.0) for k in counters)
            mix[name] = n
            named += n
        mix["other"] = max(0.0, c["SQ_INSTS_VALU"] - named)
        for name, n in mix.items():
            cost = sum(costs[p] for p in CLASSES[name][1]) / len(CLASSES[name][1])
            model_ns += n * cost / SIMDS
        c["valu_mix"] = mix
        c["valu_issue_model_ms"] = model_ns * 1e-6
irs = pmc.get("path_kernel", {}).get("_launches", 0) + pmc.get("path_pair_kernel", {}).get("_launches", 0)
for c in pmc.values():                                   # a kernel launched more than once per IR (binning: diffuse + images)
    if "hbm_bytes_per_launch" in c and irs:
        c["hbm_bytes_per_ir"] = c["hbm_bytes_per_launch"] * c.get("_launches", irs) / irs
        c["hbm_bytes_calibrated_per_ir"] = c["hbm_bytes_calibrated_per_launch"] * c.get("_launches", irs) / irs
step_kernels = [k for k in pmc if k != "attenuate_kernel"]          # the attenuate probe runs outside the timed steps
valu_per_ir = sum(pmc[k].get("SQ_INSTS_VALU_all_launches", 0.0) for k in step_kernels) / irs if irs else None
json.dump({"irs_in_command": irs, "valu_wave_instructions_per_ir": valu_per_ir, "_how": "tools/profile.sh %s: rocprofv3 --kernel-trace --pmc <one set per pass> -- python3 bench.py --steps 3 --warmup 1 "
                   "--no-cpu-baseline; per-launch averages; hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE counts "
                   "128-B requests as 64 B: MI355X_MICROARCH.md HBM section; upper bound for gather-heavy kernels)" % tag,
           "kernels": pmc}, open(os.path.join(out, tag + "_pmc.json"), "w"), indent=1)
print(json.dumps({k: {c: round(v, 4) if isinstance(v, float) else v for c, v in d.items() if not c.startswith("SQ_INSTS_V")} for k, d in pmc.items()
                  if k in ("path_kernel", "path_pair_kernel", "path_pair_group_kernel", "shadow_kernel", "shadow_pair_kernel")}, indent=1))
