#!/bin/bash
# rocprofv3 counter passes of the step stream's OPTION paths: config 5's shard (per-robot Ib and thrust
# gain, B = 2^17) and the SURVEY 8(f-3) gain sweep (per-robot weights, B = 65 536); counters in their own runs, no trace
# domains, the program directly behind `--`. Also the kernel stats of the quad form (B = 16 384 and the B = 4 096 shape).
# usage: tools/run_profiles_options.sh <outdir>; then tools/profile_summary_options.py <outdir> r04
set -o pipefail
OUT=${1:-gpurun_out/prof_opt}
mkdir -p "$OUT"
export TMPDIR=/tmp
BASE="python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition --steps 100 --warmup 100"
C5="$BASE --monte-carlo --batch 131072"
F3="$BASE --gain-sweep"
for cfg in c5 f3; do
  if [ $cfg = c5 ]; then CMD=$C5; else CMD=$F3; fi
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/${cfg}_fetch" -o pmc -- $CMD > "$OUT/${cfg}_fetch.log" 2>&1 || exit 1
  rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/${cfg}_write" -o pmc -- $CMD > "$OUT/${cfg}_write.log" 2>&1 || exit 1
done
# the quad form (one robot per lane quad), the shape small batches take: kernel stats + SQ counters at B = 16 384
Q="python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition --steps 100 --warmup 100 --batch 16384"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktq" -o kt -- $Q > "$OUT/ktq.log" 2>&1 || exit 1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d "$OUT/q_sq1" -o pmc -- $Q > "$OUT/q_sq1.log" 2>&1 || exit 1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$OUT/q_sq2" -o pmc -- $Q > "$OUT/q_sq2.log" 2>&1 || exit 1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/q_fetch" -o pmc -- $Q > "$OUT/q_fetch.log" 2>&1 || exit 1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/q_write" -o pmc -- $Q > "$OUT/q_write.log" 2>&1 || exit 1
find "$OUT" -name "*.csv" | head -30
