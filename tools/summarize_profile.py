#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/, scratch) into the small files kept under profiles/.

  python tools/summarize_profile.py <round tag> <kernel_stats.csv> [<fetch_counter_collection.csv> <write_counter_collection.csv>]

Writes profiles/<tag>_kernel_stats.csv (the rocprofv3 --kernel-trace --stats summary, library kernels
first) and profiles/<tag>_hbm_traffic.csv (per-kernel average FETCH_SIZE / WRITE_SIZE from separate --pmc
passes, with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide coalesced
read, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is taken as is)."""
import collections
import csv
import os
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0].replace("vbzhip::", "")
    return n[:90]


SIDE = " [second launch group of per-read routing: small grids]"


class GridClasses:
    """Per-read routing (vbz_api.hip) launches the large-read kernels a second time per call with small grids -- for the long
    reads of the batch, usually none.  Those launches bear the names of the main ones (same templates); averaged in, they
    would halve every per-launch figure.  A launch whose grid is below half of the kernel's largest is reported on a
    line of its own."""

    def __init__(self):
        self.max_grid = collections.defaultdict(int)

    def see(self, name, grid):
        self.max_grid[name] = max(self.max_grid[name], grid)

    def key(self, name, grid):
        return short(name) + (SIDE if "vbzhip" in name and 2 * grid < self.max_grid[name] else "")


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    os.makedirs(out, exist_ok=True)
    trace = stats.replace("kernel_stats.csv", "kernel_trace.csv")
    if os.path.exists(trace):   # per-launch records: the same figures as rocprofv3's --stats summary, main and side launches apart
        launches = list(csv.DictReader(open(trace)))
        gc = GridClasses()
        for r in launches:
            gc.see(r["Kernel_Name"], int(r["Grid_Size_X"]))
        agg = collections.defaultdict(list)
        for r in launches:
            agg[gc.key(r["Kernel_Name"], int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        total_all = sum(sum(v) for v in agg.values())
        rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
        with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ms", "avg_ms", "percent", "min_ms", "max_ms"])
            for k, v in rows[:24]:
                w.writerow([k, len(v), "%.3f" % (sum(v) / 1e6), "%.4f" % (sum(v) / len(v) / 1e6), "%.2f" % (100.0 * sum(v) / total_all),
                            "%.4f" % (min(v) / 1e6), "%.4f" % (max(v) / 1e6)])
    else:
        rows = list(csv.DictReader(open(stats)))
        rows.sort(key=lambda r: (0 if "vbzhip" in r["Name"] else 1, -float(r["TotalDurationNs"])))
        with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ms", "avg_ms", "percent", "min_ms", "max_ms"])
            for r in rows[:16]:
                w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6), "%.4f" % (float(r["AverageNs"]) / 1e6),
                            r["Percentage"], "%.4f" % (float(r["MinNs"]) / 1e6), "%.4f" % (float(r["MaxNs"]) / 1e6)])
    if len(sys.argv) >= 5:
        agg = collections.defaultdict(lambda: {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
        for path in sys.argv[3:5]:
            recs = list(csv.DictReader(open(path)))
            gc = GridClasses()
            for r in recs:
                gc.see(r["Kernel_Name"], int(r["Grid_Size"]))
            for r in recs:
                a = agg[gc.key(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        with open(os.path.join(out, tag + "_hbm_traffic.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "launches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "read_MB_corrected(2x)", "write_MB", "hbm_MB_per_launch"])
            for k, v in sorted(agg.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"][1] + kv[1]["WRITE_SIZE"][1])):
                if "svb_" not in k and "zstd_" not in k and "fast_" not in k and "elementwise" not in k and "seg_plan" not in k and "plan_scratch" not in k:
                    continue
                fn, fs = v["FETCH_SIZE"]
                wn, ws = v["WRITE_SIZE"]
                fa = fs / max(fn, 1)
                wa = ws / max(wn, 1)
                w.writerow([k, max(fn, wn), "%.1f" % fa, "%.1f" % wa, "%.1f" % (2 * fa * 1024 / 1e6), "%.1f" % (wa * 1024 / 1e6),
                            "%.1f" % ((2 * fa + wa) * 1024 / 1e6)])
    if len(sys.argv) >= 6:
        # SQ counters of further --pmc passes: per-kernel averages per launch, one column per counter
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
        names = []
        for path in sys.argv[5:]:
            recs = list(csv.DictReader(open(path)))
            gc = GridClasses()
            for r in recs:
                gc.see(r["Kernel_Name"], int(r["Grid_Size"]))
            for r in recs:
                k = gc.key(r["Kernel_Name"], int(r["Grid_Size"]))
                if "svb_" not in k and "zstd_" not in k and "fast_" not in k and "vbz_" not in k:
                    continue
                a = agg[k][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                if r["Counter_Name"] not in names:
                    names.append(r["Counter_Name"])
        with open(os.path.join(out, tag + "_sq_counters.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "launches"] + names)
            for k, v in sorted(agg.items()):
                w.writerow([k, max(x[0] for x in v.values())] + ["%.4g" % (v[c][1] / max(v[c][0], 1)) if c in v else "" for c in names])


if __name__ == "__main__":
    main()
