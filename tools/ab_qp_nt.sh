#!/bin/bash
# Round 5 experiment: non-temporal hint on the p5f QP kernel's row accesses and / or its stream blocks (UMPC_QP_NT_KINDS,
# asmqp.fmt): ms per tick and HBM bytes per robot-tick of BASELINE configs[3]. usage (GPU box): tools/ab_qp_nt.sh <outdir> <variants...>
set -o pipefail
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition --workload p5f --steps 200 --warmup 50"
for v in "$@" "$@"; do
  if [ $v = base ]; then unset UMPC_LIB; else export UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_$v.so; fi
  $CMD > "$OUT/time_$v.$RANDOM.json" 2>"$OUT/err.log" || exit 1
done
for v in "$@"; do
  if [ $v = base ]; then unset UMPC_LIB; else export UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_$v.so; fi
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch_$v" -o pmc -- $CMD --steps 20 --warmup 5 > "$OUT/fetch_$v.log" 2>&1 || exit 1
  rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/write_$v" -o pmc -- $CMD --steps 20 --warmup 5 > "$OUT/write_$v.log" 2>&1 || exit 1
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, json, os, sys
import numpy as np
out = sys.argv[1]
B = 16384
print("variant        ms/tick (200 ticks, two runs)   QP kernel: FETCH B/robot-tick (x2 corrected)   WRITE B/robot-tick   ratio to 3380 B")
for v in sys.argv[2:]:
    ts = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(os.path.join(out, "time_%s.*.json" % v)))]
    def mean(kind):
        f = glob.glob(os.path.join(out, "%s_%s" % (kind, v), "**", "pmc_counter_collection.csv"), recursive=True)[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "bqp_fixed_p5f10_asm_kernel" in r["Kernel_Name"]]
        return float(np.mean(vals[5:]))
    f, w = 2 * mean("fetch") * 1024 / B, mean("write") * 1024 / B
    print("%-13s  %s   %8.0f   %8.0f   %.3f" % (v, " ".join("%.4f" % t["ms_per_step"] for t in ts), f, w, (f + w) / 3380))
PY
