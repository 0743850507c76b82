#!/bin/bash
# Round 5 experiment: cache-policy bits on the workspace accesses of the step kernel (UMPC_ASM_WS_ST / UMPC_ASM_WS_LD): does any
# of them keep the parked rows from being written back to HBM every step? usage (GPU box): tools/ab_ws.sh <outdir> <variants...>
set -o pipefail
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition"
for v in "$@"; do
  if [ $v = base ]; then unset UMPC_LIB; else export UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_$v.so; fi
  $CMD > "$OUT/time_$v.json" 2>"$OUT/err.log" || exit 1
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch_$v" -o pmc -- $CMD > "$OUT/fetch_$v.log" 2>&1 || exit 1
  rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/write_$v" -o pmc -- $CMD > "$OUT/write_$v.log" 2>&1 || exit 1
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
B, K = 65536, 500
print("variant        ms/step (K=500)   FETCH B/robot-step (x2 corrected)   WRITE B/robot-step   ratio to 1208 B")
for v in sys.argv[2:]:
    t = json.loads(open(os.path.join(out, "time_%s.json" % v)).read().strip().splitlines()[-1])
    def last(kind):
        f = glob.glob(os.path.join(out, "%s_%s" % (kind, v), "**", "pmc_counter_collection.csv"), recursive=True)[0]
        return [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "umpc_rollout_asm_kernel" in r["Kernel_Name"]][-1]
    f, w = 2 * last("fetch") * 1024 / (B * K), last("write") * 1024 / (B * K)
    print("%-13s  %.4f   %8.0f   %8.0f   %.3f" % (v, t["ms_per_step"], f, w, (f + w) / 1208))
PY
