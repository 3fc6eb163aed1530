#!/usr/bin/env python3
"""Developer tool: merges the PMC summaries of tools/profile.sh <tag> and tools/profile.sh <tag>pairs RVB_PATH_LANES=2 into
profiles/<tag>_pmc_n1.json (what bench.py reads: the default command's kernels plus the two-lane path kernel the pipeline launches)
and copies the kernel-stats tables beside it.      python tools/merge_pmc.py r04c"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out, prof = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
new = json.load(open(os.path.join(out, tag + "_pmc.json")))
pairs = json.load(open(os.path.join(out, tag + "pairs_pmc.json")))
new["kernels"]["path_pair_kernel"] = pairs["kernels"]["path_pair_group_kernel"]
new["kernels"]["path_pair_kernel"]["_from"] = ("profiles/%s_pmc_path_pairs_n1.json (tools/profile.sh %spairs RVB_PATH_LANES=2: path_pair_group_kernel, "
                                               "the kernel the pipeline launches)" % (tag, tag))
new["_calibration"] = ("hbm_bytes_calibrated_*: FETCH_SIZE x 2 for the streaming kernels, x 1 for the gather kernels (tools/fetch_calibration.hip, "
                       "profiles/r04_fetch_calibration_n1.txt); hbm_bytes_per_*: x 2 throughout (upper bound)")
json.dump(new, open(os.path.join(prof, tag + "_pmc_n1.json"), "w"), indent=1)
shutil.copy(os.path.join(out, tag + "pairs_pmc.json"), os.path.join(prof, tag + "_pmc_path_pairs_n1.json"))
shutil.copy(os.path.join(out, tag + "_kernel_stats.csv"), os.path.join(prof, tag + "_kernel_stats_n1.csv"))
shutil.copy(os.path.join(out, tag + "_kernel_stats_default.csv"), os.path.join(prof, tag + "_kernel_stats_default_command_n1.csv"))
shutil.copy(os.path.join(out, tag + "pairs_kernel_stats.csv"), os.path.join(prof, tag + "_kernel_stats_path_pairs_n1.csv"))
for k in ("path_kernel", "path_pair_kernel", "shadow_pair_kernel"):
    v = new["kernels"][k]
    print(k, "VALU %.1f M" % (v["SQ_INSTS_VALU"] / 1e6), "issue model %.3f ms" % v.get("valu_issue_model_ms", 0.0))
