#!/bin/bash
# rocprofv3 passes of the headline bench (run on the GPU box through gpurun); summaries are copied into profiles/.
# usage: tools/run_profiles.sh <outdir>
set -o pipefail
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD500="python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition"
CMD20="python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt500" -o kt -- $CMD500 > "$OUT/kt500.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt20" -o kt -- $CMD20 > "$OUT/kt20.log" 2>&1 || exit 1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch" -o pmc -- $CMD500 > "$OUT/fetch.log" 2>&1 || exit 1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/write" -o pmc -- $CMD500 > "$OUT/write.log" 2>&1 || exit 1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d "$OUT/sq1" -o pmc -- $CMD500 --steps 100 --warmup 100 > "$OUT/sq1.log" 2>&1 || exit 1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE -d "$OUT/sq2" -o pmc -- $CMD500 --steps 100 --warmup 100 > "$OUT/sq2.log" 2>&1 || exit 1
# the two small-batch configurations (assembly ADMM loops of asmgen64.py / asmqp.py): kernel stats only
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktc2" -o kt -- python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition --dtype f64 --batch 4096 --plant euler --steps 20 --warmup 5 > "$OUT/ktc2.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktc4" -o kt -- python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition --workload p5f --steps 20 --warmup 5 > "$OUT/ktc4.log" 2>&1 || exit 1
find "$OUT" -name "*.csv" | head -40
