#!/usr/bin/env python3
"""Developer tool: HBM bytes per training step from the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh.
   python tools/traffic_from_pmc.py gpurun_out/profiles_<tag> <steps run under the profiler> <entry of profiles/traffic.json> [note]
Per kernel: mean FETCH_SIZE (KB, x2 on gfx950 - MI355X_MICROARCH.md, HBM section) + mean WRITE_SIZE (KB), times the launches
per step (launch count / steps), summed over the kernels of the step.  Rewrites that entry of profiles/traffic.json."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"(.{60}) (\S+)\s+n=\s*(\d+) mean=(\S+)", line)
        if m and m.group(2) == counter:
            out[m.group(1).strip()] = (int(m.group(3)), float(m.group(4)))
    return out


def short(name):
    name = re.sub(r"^(void )?dfn::", "", name)
    return name.split("(")[0]


def main():
    d, steps, entry = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    fetch, write = {}, {}
    for f in sorted(os.listdir(d)):
        if f.startswith("pmc_"):
            fetch.update(parse(os.path.join(d, f), "FETCH_SIZE"))
            write.update(parse(os.path.join(d, f), "WRITE_SIZE"))
    per, total = {}, 0.0
    for k in sorted(set(fetch) | set(write)):
        n = max(fetch.get(k, (0, 0))[0], write.get(k, (0, 0))[0])
        launches = n / steps
        if launches < 0.99:          # set-up kernels (weight packing, fills of the first step): not part of a step
            continue
        fk, wk = fetch.get(k, (0, 0.0))[1], write.get(k, (0, 0.0))[1]
        b = (2.0 * fk + wk) * 1024.0 * launches
        key = short(k)
        if key in per:               # template instances that share the 60-character prefix
            key = key + "#" + str(len(per))
        per[key] = {"launches_per_step": round(launches, 2), "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes_per_step": b}
        total += b
    per = dict(sorted(per.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"]))
    path = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(path))
    t[entry] = {"kernel": "whole training step (all launches)", "hbm_bytes_per_launch": total, "per_kernel": per, "note": note}
    json.dump(t, open(path, "w"), indent=1)
    print(f"{entry}: {total / 1e9:.3f} GB per step")
    for k, v in list(per.items())[:8]:
        print(f"  {k:40s} {v['hbm_bytes_per_step'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
