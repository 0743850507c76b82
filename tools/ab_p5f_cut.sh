#!/bin/bash
# Round 5: the p5f loop with each horizon chain cut in two halves on two wavefronts (qpstruct.bisect_ordering, asmqp.LoopSplit)
# against round 4's split by whole components (variant `nocut`, built with UMPC_QP_ORDERING=minfill UMPC_QP_TREE_SPLIT=0; the
# front end must then order the same way at run time: the specialisation is looked up by the hash of the structure's tables).
# ms per tick of BASELINE configs[3], alternating on one box. usage (GPU box): tools/ab_p5f_cut.sh <outdir> [reps]
set -o pipefail
OUT=$1; REPS=${2:-3}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition --workload p5f"
for rep in $(seq $REPS); do
  for v in nocut cut; do
    if [ $v = cut ]; then unset UMPC_LIB UMPC_QP_ORDERING; else export UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_nocut.so UMPC_QP_ORDERING=minfill; fi
    $CMD --steps 200 --warmup 50 > "$OUT/t200_${v}_$rep.json" 2>"$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
    $CMD --steps 20 --warmup 5 > "$OUT/t20_${v}_$rep.json" 2>"$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
    python3 - "$OUT" $v $rep <<'PY'
import json, sys
out, v, rep = sys.argv[1:]
a = json.loads(open("%s/t200_%s_%s.json" % (out, v, rep)).read().strip().splitlines()[-1])
b = json.loads(open("%s/t20_%s_%s.json" % (out, v, rep)).read().strip().splitlines()[-1])
print("%-6s rep%s K=200 ms/tick %.5f   K=20 W=5 first pass %.5f" % (
    v, rep, a["ms_per_step"], b["ms_per_step"]))
PY
  done
done
