#!/usr/bin/env python3
"""profiles/<tag>_pmc.json from a gpu_session.sh pmc step: per-kernel HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE,
KiB -> bytes; corrections per MI355X_MICROARCH.md section HBM) plus the raw counter averages.
usage: tools/make_pmc_json.py gpurun_out/<session>/pmc_summary_<workload>.json profiles/<tag>_pmc.json"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
summ = json.load(open(src))
hbm, counters = {}, {}
for k, e in summ.items():
    if not k.startswith("k_") or "fetch_bytes_x2" not in e or "write_bytes" not in e:
        continue  # the library's own kernels only (torch's fill / copy kernels of the bench are not part of a build)
    hbm[k] = e["fetch_bytes_x2"] + e["write_bytes"]
    counters[k] = {c: v for c, v in e.items() if not c.endswith("_n")}
json.dump({
    "source": "rocprofv3 --pmc passes (one counter group per pass) of `python bench.py --steps 5 --warmup 1 "
              "--no-cpu-baseline --no-cfg4-baseline --profile-reps 1 [--workload W]` (tools/gpu_session.sh pmc)",
    "corrections": "FETCH_SIZE and WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B)",
    "hbm_bytes_per_launch": hbm, "counters": counters}, open(dst, "w"), indent=1, sort_keys=True)
print({k: round(v / 1e6, 1) for k, v in hbm.items()}, "MB per launch")
