#!/usr/bin/env python3
"""Developer tool: what the GPU ran when in the bench pipeline — from the kernel trace of
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o run -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras
Prints, for the last `steps` impulse responses: when each path kernel started and how long it ran, how much of the time 0 / 1 / 2 / more
path kernels were on the GPU, and how far the other stages' kernels stretch against their durations with the GPU to themselves.
    python tools/pipeline_timeline.py <dir>/run_kernel_trace.csv [steps] [solo_kernel_stats.csv]"""
import csv
import sys
from collections import Counter, defaultdict

NAMES = ["path_pair_group_kernel", "path_pair_kernel", "path_kernel", "shadow_pair_kernel", "shadow_kernel", "image_plan_kernel", "image_check_kernel", "ordered_sum_hrtf_kernel", "ordered_sum_kernel", "bin_keys_hrtf_kernel", "bin_keys_kernel",
         "bin_starts_kernel", "radix_sort_onesweep_iteration", "radix_sort_onesweep_global_offsets", "histogram_fast_kernel",
         "histogram_transpose_kernel", "time_range_kernel", "fillBuffer", "copyBuffer"]


def short(name):
    for n in NAMES:
        if n in name:
            return n
    return name[:40]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    events = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Stream_Id"]) for r in rows)
    paths = [e for e in events if e[2] in ("path_pair_group_kernel", "path_pair_kernel", "path_kernel")]
    # the timed region's path kernels are the two-trace group launches (grid > one trace's waves); bench.py traces single IRs before and
    # after it (solo passes, the check of the last histogram), which do not belong to the region
    grid = {(int(r["Start_Timestamp"]), r["Kernel_Name"]): int(r["Grid_Size_X"]) for r in rows}
    widest = max((grid.get((e[0], n), 0) for e in paths for n in [next(r["Kernel_Name"] for r in rows if int(r["Start_Timestamp"]) == e[0])]), default=0)
    groups = [e for e in paths if grid.get((e[0], next(r["Kernel_Name"] for r in rows if int(r["Start_Timestamp"]) == e[0])), 0) == widest]
    if len(groups) >= steps // 2 and widest:
        region = groups[-(steps // 2):]
        t1 = region[-1][1] + 5_000_000                    # the last group's shadow / binning stages follow its path kernel within a few ms
        events = [e for e in events if e[0] <= t1]
        t1 = max(e[1] for e in events)
    else:
        region = paths[-steps:]
        t1 = max(e[1] for e in events)
    t0 = region[0][0]
    print("timed region: %d path kernels for %d IRs, %.2f ms = %.3f ms per IR" % (len(region), steps, (t1 - t0) / 1e6, (t1 - t0) / 1e6 / steps))
    print("path kernels (start ms, duration ms, stream):")
    for s, e, n, st in region:
        print("  %8.2f %6.2f  stream %s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, st, n))
    # residency of path kernels over the region
    points = []
    for s, e, n, _ in events:
        if e <= t0:
            continue
        points.append((max(s, t0), 1, n))
        points.append((e, -1, n))
    points.sort()
    running, last = Counter(), t0
    by_paths, idle = Counter(), 0
    for t, d, n in points:
        if t > last:
            k = running["path_pair_group_kernel"] + running["path_pair_kernel"] + running["path_kernel"]
            by_paths[min(k, 3)] += t - last
            if sum(running.values()) == 0:
                idle += t - last
        running[n] += d
        last = t
    total = float(t1 - t0)
    print("share of the region with 0 / 1 / 2 / 3+ path kernels on the GPU: " + " / ".join("%.0f %%" % (100 * by_paths[k] / total) for k in range(4))
          + "; nothing at all on the GPU %.1f %%" % (100 * idle / total))
    # stretch of the other kernels
    solo = {}
    if len(sys.argv) > 3:
        for r in csv.DictReader(open(sys.argv[3])):
            solo.setdefault(short(r["Name"]), []).append(float(r["AverageNs"]) / 1e6)
    durations = defaultdict(list)
    for s, e, n, _ in events:
        if s >= t0:
            durations[n].append((e - s) / 1e6)
    print("kernel durations in the region (ms): mean / max" + (" / alone on the GPU" if solo else ""))
    for n in NAMES:
        if durations.get(n):
            d = durations[n]
            line = "  %-36s %6.2f / %6.2f" % (n, sum(d) / len(d), max(d))
            if n in solo:
                line += " / %s" % ", ".join("%.2f" % v for v in sorted(set(round(x, 2) for x in solo[n])))
            print(line + "   (%d launches)" % len(d))


if __name__ == "__main__":
    main()
