#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of tools/profile.sh into the files committed under profiles/.

    python tools/pmc_summary.py TAG        # reads gpurun_out/prof_TAG/<run>/, writes gpurun_out/prof_TAG/out/TAG_*
    python tools/pmc_summary.py --record-build FILE   # what tools/profile.sh runs FIRST: the build id and the step
                                                      # kernels' instruction-stream digests of the library the counters
                                                      # are about to be collected on

(tools/profile.sh runs it on the GPU box and then deletes the raw traces: gpurun brings back at most 64 MiB;
copy gpurun_out/prof_TAG/out/* into profiles/ to commit them.)

* kernel stats  (``*_kernel_stats.csv`` of the --kernel-trace --stats runs) are copied as they are;
* counter runs  (``*_counter_collection.csv``): per kernel and counter the mean over the step-kernel dispatches of
  the timed part (burn-in / reset / policy kernels are listed separately), plus the derived figures
  DESIGN.md quotes: VALU issue utilisation, active-lane utilisation, wave occupancy, LDS bank-conflict share,
  HBM-side bytes (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note, WRITE_SIZE exact);
* ``traffic.json`` gets / keeps one entry per workload with measured bytes per launch (read by bench.py).
"""
import csv
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
STEP_KERNELS = ("pz::step_pair_kernel", "pz::step_kernel", "pz::rollout_pair_kernel")


def build_record():
    """{build_id, kernel_digests} of the library on disk (tools/kernel_digest.py)."""
    sys.path.insert(0, str(REPO / "pika-zoo_amd"))
    sys.path.insert(0, str(REPO / "tools"))
    import kernel_digest
    from pikazoo_amd import _native

    return {"build_id": _native.build_id(),
            "kernel_digests": {k: d for k, (d, _) in kernel_digest.kernels(_native.LIB_PATH).items() if k.startswith(STEP_KERNELS)}}


def short(name):
    return name.split("(")[0].replace("void ", "")


def counter_means(csv_path, trace_path=None):
    """{kernel: {counter: mean value per dispatch}} and {kernel: dispatches} of one counter_collection.csv."""
    sums, counts = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    meta = {}
    with open(csv_path) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            c = row["Counter_Name"]
            sums[k][c] += float(row["Counter_Value"])
            counts[k][c] += 1
            meta[k] = {"grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]),
                       "vgpr": int(row["VGPR_Count"]), "sgpr": int(row["SGPR_Count"]),
                       "lds": int(row["LDS_Block_Size"])}
    means = {k: {c: sums[k][c] / counts[k][c] for c in sums[k]} for k in sums}
    disp = {k: max(counts[k].values()) for k in counts}
    return means, disp, meta


def kernel_avg_ns(trace_csv):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(trace_csv) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            tot[k] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}, dict(cnt)


def sample_rows(csv_path, out_path, per_kernel=24):
    """The first `per_kernel` rows of every step kernel of a counter_collection.csv, as evidence."""
    seen = defaultdict(int)
    with open(csv_path) as f, open(out_path, "w", newline="") as g:
        r = csv.reader(f)
        w = csv.writer(g)
        header = next(r)
        w.writerow(header)
        ki = header.index("Kernel_Name")
        for row in r:
            k = short(row[ki])
            if k.startswith(STEP_KERNELS) and seen[k] < per_kernel:
                seen[k] += 1
                w.writerow(row)


def stats_by_grid(trace_csv, out_path):
    """Kernel durations of a --kernel-trace run grouped by (kernel, grid size): one bench run times several batch
    sizes through the same kernel, which the tool's own --stats table lumps together."""
    acc = defaultdict(list)
    with open(trace_csv) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if k.startswith(STEP_KERNELS):
                acc[(k, int(row["Grid_Size_X"]), int(row["Workgroup_Size_X"]))].append(
                    int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    with open(out_path, "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["Kernel", "Grid_Size_X", "Workgroup_Size_X", "Games", "Calls", "AverageNs", "MedianNs", "MinNs", "MaxNs"])
        for (k, grid, wg), v in sorted(acc.items()):
            v.sort()
            w.writerow([k, grid, wg, grid // wg * 64, len(v), round(sum(v) / len(v), 1), v[len(v) // 2], v[0], v[-1]])


def calibration(src, dst, tag):
    """Known-byte kernels (tools/calib_traffic.hip) under --pmc FETCH_SIZE / WRITE_SIZE: counter value x 1024 over the
    bytes the kernel moved, per access shape and buffer size.  -> TAG_calibration.json"""
    import re

    out = {}
    for run, counter in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
        cc = find(src / run, "_counter_collection.csv") if (src / run).is_dir() else None
        log = src / f"{run}.log"
        if cc is None or not log.exists():
            continue
        known = {}  # (kernel, lanes) -> (where, bytes)
        for m in re.finditer(r"^calib (\S+) (\S+) grid=(\d+) bytes=(\d+)", log.read_text(), re.M):
            known[(m.group(2), int(m.group(3)) * 64)] = (m.group(1), int(m.group(4)))
        acc = defaultdict(list)
        with open(cc) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter:
                    acc[(short(row["Kernel_Name"]).split("<")[0], int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
        for (k, lanes), vals in sorted(acc.items()):
            if (k, lanes) not in known:
                continue
            where, nbytes = known[(k, lanes)]
            vals = vals[1:] or vals  # the first launch of a shape also pays for cold pages
            mean = sum(vals) / len(vals)
            out.setdefault(counter, {})[f"{k}@{where}"] = {
                "known_bytes": nbytes, "counter_kb_mean": mean, "launches": len(vals),
                "counter_bytes_over_known_bytes": mean * 1024.0 / nbytes}
    if out:
        (dst / f"{tag}_calibration.json").write_text(json.dumps(out, indent=1))
        print(json.dumps(out, indent=1))
    return out


def find(run_dir, suffix):
    hits = sorted(Path(run_dir).rglob(f"*{suffix}"))
    return hits[0] if hits else None


def main():
    tag = sys.argv[1]
    src = REPO / "gpurun_out" / f"prof_{tag}"
    dst = src / "out"
    dst.mkdir(exist_ok=True)
    summary = {}
    calibration(src, dst, tag)
    for run in sorted(p for p in src.iterdir() if p.is_dir() and p.name != "out"):
        name = run.name
        if name.startswith("calib_"):
            continue
        stats = find(run, "_kernel_stats.csv")
        if stats is not None:
            shutil.copy(stats, dst / f"{tag}_{name}_kernel_stats.csv")
            tr = find(run, "_kernel_trace.csv")
            if tr is not None:
                stats_by_grid(tr, dst / f"{tag}_{name}_step_kernels_by_grid.csv")
        cc = find(run, "_counter_collection.csv")
        if cc is None:
            continue
        means, disp, meta = counter_means(cc)
        sample_rows(cc, dst / f"{tag}_{name}_counter_sample.csv")
        trace = find(run, "_kernel_trace.csv")
        avg_ns, _ = kernel_avg_ns(trace) if trace else ({}, {})
        entry = {}
        for k in means:
            if not k.startswith(STEP_KERNELS):
                continue
            entry[k] = {"dispatches": disp[k], "avg_ns_under_pmc": avg_ns.get(k), **meta[k], "counters": means[k]}
        summary[name] = entry
    # derived figures per workload prefix (hh / cfg3 / big)
    derived = {}
    for prefix in sorted({n.rsplit("_", 1)[0] for n in summary}):
        d = {}
        fetch = summary.get(f"{prefix}_fetch", {})
        write = summary.get(f"{prefix}_write", {})
        for k, e in fetch.items():
            d.setdefault(k, {})["fetch_size_kb_raw"] = e["counters"].get("FETCH_SIZE")
        for k, e in write.items():
            d.setdefault(k, {})["write_size_kb"] = e["counters"].get("WRITE_SIZE")
        for k, v in d.items():
            if v.get("fetch_size_kb_raw") is not None and v.get("write_size_kb") is not None:
                # gfx950: FETCH_SIZE reports 1/2 of coalesced read bytes (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact
                v["hbm_bytes_per_launch"] = (2.0 * v["fetch_size_kb_raw"] + v["write_size_kb"]) * 1024.0
        sqa = summary.get(f"{prefix}_sqa", {})
        sqb = summary.get(f"{prefix}_sqb", {})
        for k, e in sqa.items():
            c = e["counters"]
            v = d.setdefault(k, {})
            waves, wc = c.get("SQ_WAVES"), c.get("SQ_WAVE_CYCLES")
            if waves and wc:
                v["waves_per_launch"] = waves
                v["wave_cycles_per_wave_x4"] = 4.0 * wc / waves           # SQ_WAVE_CYCLES counts quad-cycles
            if c.get("SQ_BUSY_CYCLES") and wc:
                v["avg_waves_resident_per_busy_SQ_cycle"] = wc / c["SQ_BUSY_CYCLES"]
            if c.get("SQ_ACTIVE_INST_VALU") and wc:
                v["valu_issue_share_of_wave_cycles"] = c["SQ_ACTIVE_INST_VALU"] / wc
            if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
                # thread-cycles / (instruction-cycles x 64 lanes): share of lanes active in issued VALU work
                v["valu_active_lane_share"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
            if c.get("SQ_INSTS_VALU") and waves:
                v["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / waves
            for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
                if c.get(name) and wc:
                    v[name.lower() + "_share"] = c[name] / wc
        for k, e in sqb.items():
            c = e["counters"]
            v = d.setdefault(k, {})
            if c.get("SQ_LDS_IDX_ACTIVE"):
                v["lds_bank_conflict_share"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
            for name in ("SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if name in c:
                    v[name.lower() + "_per_launch"] = c[name]
        derived[prefix] = d
    out = {"tag": tag, "derived": derived, "raw_means": summary}
    # which build the counters were taken on, and the instruction-stream digest of every step kernel of that build:
    # bench.py accepts these figures on another build only for a kernel whose digest is unchanged (tools/kernel_digest.py).
    # The record was written by profile.sh BEFORE its first rocprofv3 run (build_at_collection.json); a library rebuilt
    # between collection and this summary would otherwise stamp the counters with the wrong build.
    recorded = src / "build_at_collection.json"
    try:
        now = build_record()
        if recorded.exists():
            then = json.loads(recorded.read_text())
            if then["build_id"] != now["build_id"]:
                raise SystemExit(f"the library was rebuilt since the counters were collected ({then['build_id']} -> "
                                 f"{now['build_id']}): refusing to stamp them with another build")
            out.update(then)
        else:
            print("(no build_at_collection.json beside the traces: stamping the library on disk)")
            out.update(now)
    except SystemExit:
        raise
    except Exception as exc:  # noqa: BLE001
        print(f"(no build id / kernel digests: {type(exc).__name__}: {exc})")
    if summary:  # (a run of kernel-trace sections only has no counters: it must not replace a full run's summary, which
        # gpurun would then overwrite in the caller's gpurun_out/)
        (dst / f"{tag}_pmc_summary.json").write_text(json.dumps(out, indent=1))
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--record-build":
        Path(sys.argv[2]).write_text(json.dumps(build_record(), indent=1))
    else:
        main()
