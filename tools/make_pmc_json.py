#!/usr/bin/env python3
"""profiles/<tag>_pmc.json from a gpu_session.sh pmc step: per-kernel HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE,
KiB -> bytes; corrections per MI355X_MICROARCH.md section HBM) plus the raw counter averages.
usage: tools/make_pmc_json.py gpurun_out/<session> profiles/<tag>_pmc.json"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
summ = json.load(open(f"{src}/pmc_summary.json"))
names = {"k_sweep_count_masks_f32": "k_sweep_count_masks_f32", "k_sweep_count_f32": "k_sweep<COUNT>",
         "k_sweep_mfma_f32": "k_sweep_mfma_f32", "k_sweep<float, 1": "k_sweep<FILL>",
         "k_fill_masks<float": "k_fill_masks<float>", "k_bin_rows<float>": "k_bin_rows<float>",
         "k_bin_scatter<float>": "k_bin_scatter<float>", "k_bin_cells<float>": "k_bin_cells<float>",
         "k_hash<float>": "k_hash<float>", "k_reorder<float>": "k_reorder<float>", "k_row_base": "k_row_base"}
hbm, counters = {}, {}
for k, e in summ.items():
    key = next((v for p, v in names.items() if k.startswith(p)), None)
    if key is None or "fetch_bytes_x2" not in e or "write_bytes" not in e:
        continue
    hbm[key] = e["fetch_bytes_x2"] + e["write_bytes"]
    counters[key] = {c: v for c, v in e.items() if not c.endswith("_n")}
json.dump({
    "source": "rocprofv3 --pmc passes (one counter group per pass) of `python bench.py --steps 5 --warmup 1 "
              "--no-cpu-baseline --profile-reps 1` (tools/gpu_session.sh pmc)",
    "corrections": "FETCH_SIZE and WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B)",
    "hbm_bytes_per_launch": hbm, "counters": counters}, open(dst, "w"), indent=1, sort_keys=True)
print({k: round(v / 1e6, 1) for k, v in hbm.items()}, "MB per launch")
