#!/bin/bash
# rocprofv3 counter passes of the two small-batch configurations (config 4: planar p5f fp32, config 2: uprightmpc2 fp64);
# counters in their own runs, no trace domains. usage: tools/run_profiles_small.sh <outdir>; then tools/profile_summary_small.py
set -o pipefail
OUT=${1:-gpurun_out/prof_small}
mkdir -p "$OUT"
export TMPDIR=/tmp
C4="python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition --workload p5f --steps 20 --warmup 5"
C2="python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition --dtype f64 --batch 4096 --plant euler --steps 20 --warmup 5"
for cfg in c4 c2; do
  if [ $cfg = c4 ]; then CMD=$C4; else CMD=$C2; fi
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/${cfg}_fetch" -o pmc -- $CMD > "$OUT/${cfg}_fetch.log" 2>&1 || exit 1
  rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/${cfg}_write" -o pmc -- $CMD > "$OUT/${cfg}_write.log" 2>&1 || exit 1
  rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d "$OUT/${cfg}_sq1" -o pmc -- $CMD > "$OUT/${cfg}_sq1.log" 2>&1 || exit 1
  rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$OUT/${cfg}_sq2" -o pmc -- $CMD > "$OUT/${cfg}_sq2.log" 2>&1 || exit 1
done
find "$OUT" -name "*.csv" | head -20
