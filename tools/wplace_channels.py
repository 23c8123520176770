#!/usr/bin/env python3
"""Per-dispatch, per-L2-channel view of a rocprofv3 --output-format json counter pass over tools/bin/wplace (diagnostic):
for every `traj<0>` dispatch the duration and, per counter, mean / max over the 128 TCC instances and the hottest
channels of every XCC."""
import json
import statistics as st
import sys

d = json.load(open(sys.argv[1]))
r = d["rocprofiler-sdk-tool"][0]
cname = {c["id"]["handle"]: c["name"] for c in r["counters"]}
kname = {k["kernel_id"]: (k.get("demangled_kernel_name") or k["kernel_name"]) for k in r["kernel_symbols"]}
i = 0
for rec in r["callback_records"]["counter_collection"]:
    di = rec["dispatch_data"]["dispatch_info"]
    kn = kname[di["kernel_id"]]
    if "traj<0>" not in kn:
        continue
    i += 1
    if i % 4:  # every fourth dispatch is enough
        continue
    dur = (rec["dispatch_data"]["end_timestamp"] - rec["dispatch_data"]["start_timestamp"]) / 1e3
    vals = {}
    for x in rec["records"]:
        vals.setdefault(cname[x["counter_id"]["handle"]], []).append(x["value"])
    line = f"{kn[:12]} dispatch {di['dispatch_id']:4d} {dur:7.1f} us"
    for c, a in sorted(vals.items()):
        line += f" | {c[4:]}: mean {st.mean(a):9.0f} max {max(a):9.0f}"
    print(line)
    a = vals.get("TCC_EA0_WRREQ_STALL")
    if a and len(a) == 128:
        print("      WRREQ_STALL k/channel by XCC:", " / ".join(" ".join(f"{int(v / 1000):3d}" for v in a[x * 16:(x + 1) * 16]) for x in (0, 2, 4, 6)))
    a = vals.get("TCC_TAG_STALL")
    if a and len(a) == 128:
        print("      TAG_STALL   k/channel by XCC:", " / ".join(" ".join(f"{int(v / 1000):3d}" for v in a[x * 16:(x + 1) * 16]) for x in (0, 2, 4, 6)))
