#!/usr/bin/env python3
"""First look at the fine-row path: stage times (HIP events) of cfg2 / cfg3 with NL_ROWS = -1 / 0, checksums."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for wl in sys.argv[1:] or ["cfg2", "cfg3"]:
    for rows in ("-1", "0"):
        env = dict(os.environ, NL_ROWS=rows)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "10", "--workload", wl,
                              "--no-cpu-baseline", "--no-cfg4-baseline"], env=env, capture_output=True, text=True, timeout=300)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print(wl, rows, "FAILED", out.stdout[-2000:], out.stderr[-3000:])
            continue
        d = json.loads(lines[-1])
        st = d["roofline"]["stages_ms"]
        print(f"{wl} NL_ROWS={rows}: {d['ms_per_step']:.4f} ms/build  " + "  ".join(f"{k} {v:.4f}" for k, v in st.items()) +
              f"  pairs {d['config'].get('half_pairs_reference')} checksum {d['config'].get('list_checksum_reference')}", flush=True)
