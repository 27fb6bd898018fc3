#!/bin/bash
# tools/chunk_ab.sh: bench stage times per binning chunk size (NL_DEBUG_BIN_CHUNK), interleaved; WL=cfg2|cfg3
OUT=gpurun_out/chunk; mkdir -p $OUT
for r in 1 2; do for c in ${CHUNKS:-4096 8192 16384 32768}; do
  NL_DEBUG_BIN_CHUNK=$c timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-cfg4-baseline --workload ${WL:-cfg2} > $OUT/b_${c}_$r.log 2>&1
  python - $OUT/b_${c}_$r.log "chunk $c" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
d = json.loads(l[-1]); st = d["roofline"]["stages_ms"]
print(f"{sys.argv[2]:12s} {d['ms_per_step']:.4f} ms/build hash {st['hash']:.4f} reorder {st['reorder']:.4f} count {st['count']:.4f} fill {st['fill']:.4f} {d['config'].get('half_pairs_reference')}")
PY
done; done
