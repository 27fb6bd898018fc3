#!/usr/bin/env python3
"""CPU rehearsal of the 8-rank slab path (gloo, per-rank build emulated by the oracle): decomposition, ghost exchange
and ownership rule at the world size the driver's scaling run uses.  The GPU suite covers 2 and 3 ranks with the HIP build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.slab_worker import run

if __name__ == "__main__":
    for case in [(20000, (12.0, 12.0, 66.5), 3.3, "float32", 181),
                 (16000, (11.0, 12.5, 60.0), 3.3, "float64", 182, [0, 9, 17])]:
        t = time.time()
        res = run(8, "oracle", case)
        print(case[:4], "->", res[0], f"{time.time() - t:.1f} s", flush=True)
        assert res[0] == "ok", res
