#!/bin/bash
# A/B of p5f library variants (tools/build_qp_variant.sh <name> under generator switches) against the shipped library on
# BASELINE configs[3], alternating on one box. usage (GPU box): tools/ab_lib_p5f.sh <outdir> <reps> base <variant>...
set -o pipefail
OUT=$1; REPS=$2; shift 2
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition --workload p5f"
for rep in $(seq $REPS); do
  for v in "$@"; do
    if [ $v = base ]; then unset UMPC_LIB; else export UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_$v.so; fi
    $CMD --steps 200 --warmup 50 > "$OUT/t200_${v}_$rep.json" 2>"$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
    $CMD --steps 20 --warmup 5 > "$OUT/t20_${v}_$rep.json" 2>"$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
    python3 - "$OUT" $v $rep <<'PY'
import json, sys
out, v, rep = sys.argv[1:]
a = json.loads(open("%s/t200_%s_%s.json" % (out, v, rep)).read().strip().splitlines()[-1])
b = json.loads(open("%s/t20_%s_%s.json" % (out, v, rep)).read().strip().splitlines()[-1])
print("%-10s rep%s K=200 ms/tick %.5f   K=20 W=5 first pass %.5f" % (v, rep, a["ms_per_step"], b["ms_per_step"]))
PY
  done
done
