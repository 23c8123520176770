#!/usr/bin/env python3
"""Per-case means of a rocprofv3 --pmc pass over tools/placement.py --replays R (diagnostic): the k-frame kernel's
dispatches come in runs of (R + 1) * launches per case; prints each run's mean duration and mean counter values.

    python tools/pmc_cases.py <dir with *_counter_collection.csv> <dispatches per case>
"""
import csv
import sys
from collections import OrderedDict, defaultdict
from pathlib import Path


def main():
    root, per_case = Path(sys.argv[1]), int(sys.argv[2])
    files = sorted(root.rglob("*counter_collection.csv"))
    assert files, f"no counter_collection.csv under {root}"
    disp = OrderedDict()
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if "step_kernel" not in row["Kernel_Name"] and "rollout_pair_kernel" not in row["Kernel_Name"]:
                    continue
                if int(row["Grid_Size"]) < 65536:
                    continue
                d = disp.setdefault(int(row["Dispatch_Id"]), {"dur": 0.0, "c": defaultdict(float)})
                d["dur"] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                d["c"][row["Counter_Name"]] += float(row["Counter_Value"])
    rows = list(disp.values())
    # the first dispatches are the env's own (allocating launch ...): the cases are the LAST complete runs
    names = sorted({k for r in rows for k in r["c"]})
    print(f"{len(rows)} k-frame dispatches, {per_case} per case; columns: case, mean us per dispatch, " + ", ".join(names))
    for i in range(0, len(rows) - per_case + 1, per_case):
        chunk = rows[i:i + per_case]
        m = len(chunk)
        print(f"case {i // per_case:2d}  {sum(r['dur'] for r in chunk) / m:9.2f} us  "
              + "  ".join(f"{sum(r['c'][k] for r in chunk) / m:14.1f}" for k in names))


if __name__ == "__main__":
    main()
