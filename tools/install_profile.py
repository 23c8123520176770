#!/usr/bin/env python3
"""Move the condensed output of one tools/profile.sh run into profiles/ (run here, after gpurun brought it back).

    python tools/install_profile.py TAG [--replace OLDTAG] [--bench gpurun_out/bench_TAG.json] [--extra ...json]

* copies gpurun_out/prof_TAG/out/TAG_* (and the bench lines, as TAG_bench_default[_extra].json) into profiles/;
* `--replace OLDTAG`: removes profiles/OLDTAG_* (the evidence of an earlier build of the same round);
* refreshes profiles/traffic.json (measured bytes per launch per workload, read by bench.py) from TAG_pmc_summary.json;
* prints the figures profiles/README.md and DESIGN.md quote, so that the prose can be checked against them.
"""
import csv
import json
import shutil
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
PROFILES = REPO / "profiles"

# traffic.json key (bench.py: traffic_key) -> (section prefix of profile.sh, kernel, games)
TRAFFIC = {
    "random_random": ("hh", "pz::step_pair_kernel<false, false, false, false>", 65536),
    "random_random_4096": ("cfg2", "pz::step_pair_kernel<false, false, false, false>", 4096),
    "cfg3": ("cfg3", "pz::step_pair_kernel<false, true, false, false>", 65536),
    "cfg3_compute": ("cfg3c", "pz::step_kernel<false, true, 0, true, 1, false, false, false>", 65536),
    "cfg3_power_hit": ("cfg3q", "pz::step_pair_kernel<false, true, false, false>", 65536),
    "cfg5": ("cfg5", "pz::step_pair_kernel<false, false, false, false>", 65536),
    "random_random_524288": ("big", "pz::step_kernel<false, false, 0, true, 0, false, false, false>", 524288),
    "random_random_int16obs": ("i16", "pz::step_pair_kernel<false, false, false, false>", 65536),
    "packed_random_random": ("pk", "pz::step_pair_kernel<false, false, true, false>", 65536),
    "packed_cfg3": ("pkcfg3", "pz::step_pair_kernel<false, true, true, false>", 65536),
    "packed_random_random_524288": ("pkbig", "pz::step_pair_kernel<false, false, true, false>", 524288),
    "packed_random_random_int16obs": ("ph", "pz::step_pair_kernel<false, false, true, false>", 65536),
    "packed_random_random_524288_int16obs": ("phbig", "pz::step_pair_kernel<false, false, true, false>", 524288),
    # the k-frame launches (k = 32: bytes per 32-frame launch; the last template argument: the PLAIN instantiation)
    "rollout_k32": ("roll", "pz::step_kernel<false, false, 2, false, 0, false, false, false>", 65536),
    "step_many_k32": ("roll", "pz::step_kernel<false, false, 3, false, 0, false, false, true>", 65536),
    "rollout_k32_p2_computer": ("roll", "pz::rollout_pair_kernel<false, true, 2, false, false, true>", 65536),
    "step_many_k32_p2_computer": ("roll", "pz::rollout_pair_kernel<false, true, 3, false, false, true>", 65536),
    "rollout_k32_int16obs": ("roll", "pz::rollout_pair_kernel<false, false, 2, false, true, true>", 65536),
}


def main():
    args = sys.argv[1:]
    tag = args[0]
    src = REPO / "gpurun_out" / f"prof_{tag}" / "out"
    if "--replace" in args:
        old = args[args.index("--replace") + 1]
        for p in PROFILES.glob(f"{old}_*"):
            p.unlink()
    for p in sorted(src.glob(f"{tag}_*")):
        shutil.copy(p, PROFILES / p.name)
    for flag, name in (("--bench", "bench_default"), ("--extra", "bench_default_extra")):
        if flag in args:
            line = Path(args[args.index(flag) + 1])
            shutil.copy(line, PROFILES / f"{tag}_{name}.json")
            # the line's verbose blocks (bench.py --configs-out; gpurun_out/... relative to the repo root)
            side = REPO / json.loads(line.read_text().strip().splitlines()[-1]).get("configs_file", "-")
            if side.is_file():
                shutil.copy(side, PROFILES / f"{tag}_{name}_configs.json")
    summary = json.loads((src / f"{tag}_pmc_summary.json").read_text())
    derived, build_id, digests = summary["derived"], summary.get("build_id"), summary.get("kernel_digests", {})
    tpath = PROFILES / "traffic.json"
    traffic = json.loads(tpath.read_text())
    for key, (prefix, kernel, games) in TRAFFIC.items():
        e = derived.get(prefix, {}).get(kernel)
        if e is None or "hbm_bytes_per_launch" not in e:
            print(f"  (no counters for {key}: section {prefix} missing)")
            continue
        traffic.setdefault(key, {}).update(hbm_bytes_per_launch=e["hbm_bytes_per_launch"],
                                           fetch_size_kb_raw=e["fetch_size_kb_raw"], write_size_kb=e["write_size_kb"],
                                           kernel=kernel, round=tag, num_envs=games, build_id=build_id,
                                           kernel_digest=digests.get(kernel))
        print(f"  traffic {key}: {e['hbm_bytes_per_launch'] / 1e6:.2f} MB = 2 x {e['fetch_size_kb_raw']:.0f} KB read + "
              f"{e['write_size_kb']:.0f} KB written")
    tpath.write_text(json.dumps(traffic, indent=1))
    for prefix in ("hh", "pk", "cfg3", "cfg3q", "cfg3c"):
        for kernel, e in derived.get(prefix, {}).items():
            if "valu_insts_per_wave" in e and ("pair" in kernel or ", 0, true, 1" in kernel):
                print(f"  {prefix} {kernel}: VALU/wave {e['valu_insts_per_wave']:.0f}, active lanes "
                      f"{e.get('valu_active_lane_share', 0):.2f}, VALU issue {e.get('valu_issue_share_of_wave_cycles', 0):.3f}, "
                      f"waiting {e.get('sq_wait_any_share', 0):.2f}, waves/busy cycle "
                      f"{e.get('avg_waves_resident_per_busy_SQ_cycle', 0):.1f}")
    for name in ("kt", "big_kt", "pkbig_kt", "phbig_kt", "roll"):
        p = src / f"{tag}_{name}_step_kernels_by_grid.csv"
        if p.exists():
            for row in csv.DictReader(p.open()):
                if int(row["Calls"]) >= 50:
                    print(f"  {name}: {row['Kernel']} games {row['Games']}: avg {float(row['AverageNs']) / 1e3:.2f} us "
                          f"({row['Calls']} calls)")
    for flag in ("--bench", "--extra"):
        if flag in args:
            d = json.loads(Path(args[args.index(flag) + 1]).read_text().strip().splitlines()[-1])
            if flag == "--bench":
                r = d["roofline"]
                print(f"  bench: build {d['build_id']}: {d['value'] / 1e9:.2f} G, {r['launch_us']:.2f} us, frac {r['frac']:.3f}, "
                      f"frac_wall {r['frac_wall']:.3f}, frac_traffic {r['frac_traffic']} (counters of {r.get('traffic_build_id')}, "
                      f"stale {r.get('traffic_stale')}), cpu {d['cpu_baseline']['value'] / 1e6:.0f} M")
                for k, v in r.items():  # the flat per-config scalars of the line
                    if k.endswith(("_launch_us", "_us_per_frame", "_us")) and k != "launch_us":
                        stem = k.rsplit("_launch_us", 1)[0].rsplit("_us_per_frame", 1)[0]
                        print(f"    {k}: {v}  frac {r.get(stem + '_frac')}  frac_traffic {r.get(stem + '_frac_traffic')}")
                print(f"    configs parity: {r.get('configs_parity_all_bit_exact')} ({r.get('configs_parity_checked')} checked)")
            else:
                side = REPO / d.get("configs_file", "-")  # the verbose blocks of the line (bench.py --configs-out)
                verbose = json.loads(side.read_text()) if side.is_file() else {}
                for k, v in verbose.get("extra", {}).items():
                    if isinstance(v, dict) and "us_per_frame" in v:
                        print(f"    {k}: {v['us_per_frame']:.2f} us/frame, {v['value'] / 1e9:.2f} G")
                    elif isinstance(v, dict) and "us_per_step" in v:
                        print(f"    {k}: {v['us_per_step']:.2f} us/step, {v['value'] / 1e9:.2f} G")
                    elif isinstance(v, dict) and "launch_us" in v:
                        print(f"    {k}: {v['launch_us']:.2f} us")
                    elif isinstance(v, dict):
                        for kk, vv in v.items():
                            print(f"    {k} {kk}: {vv['launch_us']:.2f} us, frac {vv['frac_of_8TBps']:.3f}")


if __name__ == "__main__":
    main()
