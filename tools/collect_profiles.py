"""Turn gpurun_out/prof_<tag>/ (tools/make_profiles.sh) and gpurun_out/pmc_<tag>_B*/ into the tracked files under
profiles/.  usage: python3 tools/collect_profiles.py [tag]      (default r03)"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + TAG)
DST = os.path.join(ROOT, "profiles")


def first(pattern):
    hits = glob.glob(pattern, recursive=True)
    return hits[0] if hits else None


def copy_stats(subdir, name):
    f = first(os.path.join(SRC, subdir, "**", "*kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join(DST, name))
    return f


def durations_by_grid(subdir, name):
    f = first(os.path.join(SRC, subdir, "**", "*kernel_trace.csv"))
    if not f:
        return
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X")), r.get("Workgroup_Size", r.get("Workgroup_Size_X")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(os.path.join(DST, name), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "grid_size", "workgroup_size", "calls", "avg_us", "median_us", "min_us", "max_us"])
        for (k, g, wg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            v.sort()
            w.writerow([k, g, wg, len(v), round(sum(v) / len(v), 3), round(v[len(v) // 2], 3), round(v[0], 3), round(v[-1], 3)])


def main():
    os.makedirs(DST, exist_ok=True)
    copy_stats("bench", f"{TAG}_bench_kernel_stats.csv")
    durations_by_grid("bench", f"{TAG}_bench_kernel_durations_by_grid.csv")
    for b in (512, 4096, 32768):
        copy_stats(f"kern_B{b}", f"{TAG}_kernels_B{b}_kernel_stats.csv")
    copy_stats("train", f"{TAG}_train_B512_kernel_stats.csv")
    tr = first(os.path.join(SRC, "train", "**", "*kernel_trace.csv"))
    if tr:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_step.py"), tr], capture_output=True, text=True)
        open(os.path.join(DST, f"{TAG}_train_B512_step_timeline.txt"), "w").write(out.stdout)
    for name in ("bench.json", "bench_under_rocprof.json"):
        f = os.path.join(SRC, name)
        if os.path.exists(f) and os.path.getsize(f):
            shutil.copy(f, os.path.join(DST, f"{TAG}_" + name))
    # counters
    traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/profile_kernels.py B 6; "
                         "KiB x 1024, FETCH_SIZE x 2 (gfx950: MI355X_MICROARCH.md, HBM)", "per_launch": defaultdict(dict)}
    keymap = {"gdn_dense_attn_kernel<4, 16, 0, false>": "k8", "gdn_dense_attn_kernel<4, 16, 1, false>": "k8_bf16",
              "gdn_dense_fused_kernel<4, 2, 1, 16, 0>": "fused", "gdn_dense_fused_kernel<4, 2, 1, 16, 1>": "fused_bf16",
              "gdn_dense_project_kernel<4, 1, 0>": "project", "gdn_dense_project_kernel<4, 1, 1>": "project_bf16"}
    for b in (512, 4096, 32768):
        f = os.path.join(ROOT, "gpurun_out", f"pmc_{TAG}_B{b}", "summary.json")
        if not os.path.exists(f):
            continue
        summ = json.load(open(f))
        shutil.copy(f, os.path.join(DST, f"{TAG}_sq_counters_B{b}.json"))
        for k, v in summ.items():
            short = k.split(" grid=")[0]
            if short in keymap and "hbm_read_bytes" in v:
                traffic["per_launch"][keymap[short]][str(b)] = {"fetch_bytes": v["hbm_read_bytes"], "write_bytes": v.get("hbm_write_bytes", 0.0)}
    # BASELINE configs[4]: stats + counters of the row-gather kernels at 4096 windows
    copy_stats("kern_config4_B4096", f"{TAG}_kernels_config4_B4096_kernel_stats.csv")
    f = os.path.join(ROOT, "gpurun_out", f"pmc_{TAG}_config4_B4096", "summary.json")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(DST, f"{TAG}_sq_counters_config4_B4096.json"))
        for k, v in json.load(open(f)).items():
            if "hbm_read_bytes" not in v:
                continue
            for frag, key in (("gdn_window_kernel<64, 2,", "fused_n512_d64"), ("gdn_window_kernel<64, 1,", "k8_n512_d64")):
                if frag in k:
                    traffic["per_launch"].setdefault(key, {})["4096"] = {"fetch_bytes": v["hbm_read_bytes"],
                                                                          "write_bytes": v.get("hbm_write_bytes", 0.0)}
    json.dump(traffic, open(os.path.join(DST, f"{TAG}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print("profiles updated:", sorted(os.listdir(DST)))


if __name__ == "__main__":
    main()
