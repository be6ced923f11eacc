"""rocprofv3 --pmc CSV(s) -> one JSON: per kernel, per counter, mean value per dispatch.
usage: python3 tools/pmc_summary.py out.json dir_or_csv [dir_or_csv ...]
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; FETCH_SIZE is additionally doubled here
(MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-B read requests at 64 B)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name).strip()


def main():
    out, srcs = sys.argv[1], sys.argv[2:]
    files = []
    for s in srcs:
        files += [s] if s.endswith(".csv") else glob.glob(os.path.join(s, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"]) + f" grid={row['Grid_Size']} wg={row['Workgroup_Size']}"
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                meta[k] = {"vgpr": int(row.get("VGPR_Count", 0) or 0), "agpr": int(row.get("Accum_VGPR_Count", 0) or 0),
                           "sgpr": int(row.get("SGPR_Count", 0) or 0), "lds": int(row.get("LDS_Block_Size", 0) or 0)}
    res = {}
    for k, counters in sorted(acc.items()):
        rec = dict(meta[k])
        for c, vals in sorted(counters.items()):
            v = sum(vals) / len(vals)
            if c == "FETCH_SIZE":
                rec["hbm_read_bytes"] = v * 1024 * 2
            elif c == "WRITE_SIZE":
                rec["hbm_write_bytes"] = v * 1024
            rec[c] = v
            rec.setdefault("dispatches", len(vals))
        res[k] = rec
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(f"{len(files)} csv file(s), {len(res)} kernels -> {out}")


if __name__ == "__main__":
    main()
