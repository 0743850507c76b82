#!/bin/bash
# Round 5: A/B of the non-temporal hint on the step kernel's streaming rows (UMPC_ASM_NT, asmstep.py) on ONE box:
# ms per step at K = 500 and HBM bytes per robot-step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes).
# Variants are built here in the container: UMPC_ASM_NT=1 tools/build_variant.py nt ; =ld ntld ; =st ntst
# usage (on the GPU box): tools/ab_nt.sh <outdir>
set -o pipefail
OUT=${1:-gpurun_out/nt}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --no-cpu-baseline --no-side-configs"
for v in base nt ntld ntst base nt; do
  if [ $v = base ]; then unset UMPC_LIB; else export UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_$v.so; fi
  $CMD > "$OUT/time_$v.$RANDOM.json" 2>"$OUT/err.log" || exit 1
done
for v in base nt ntld ntst; do
  if [ $v = base ]; then unset UMPC_LIB; else export UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_$v.so; fi
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch_$v" -o pmc -- $CMD --no-precondition > "$OUT/fetch_$v.log" 2>&1 || exit 1
  rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/write_$v" -o pmc -- $CMD --no-precondition > "$OUT/write_$v.log" 2>&1 || exit 1
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
B, K = 65536, 500
print("variant   ms/step first pass (K=500)   loaded      FETCH B/robot-step (x2 corrected)   WRITE B/robot-step   ratio to 1208 B")
for v in ("base", "nt", "ntld", "ntst"):
    ts = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(os.path.join(out, "time_%s.*.json" % v)))]
    def last(kind):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(glob.glob(os.path.join(out, "%s_%s" % (kind, v), "**", "pmc_counter_collection.csv"), recursive=True)[0]))
                if "umpc_rollout_asm_kernel" in r["Kernel_Name"]]
        return vals[-1]
    f, w = 2 * last("fetch") * 1024 / (B * K), last("write") * 1024 / (B * K)
    print("%-8s  %s   %s   %8.0f   %8.0f   %.3f" % (v, " ".join("%.4f" % t["ms_per_step"] for t in ts),
          " ".join("%.4f" % (t["loaded_clocks"] or {}).get("ms_per_step", float("nan")) for t in ts), f, w, (f + w) / 1208))
PY
