# config 4 (p5f): how the tick splits between the ADMM iterations and everything else, per kernel
set -o pipefail
run() { python bench.py --workload p5f --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel'])"; }
for it in 50 25 1; do echo "wave_iters$it $(run --max-iter $it)"; done
for it in 50 25 1; do echo "lane_iters$it $(UMPC_QP_KERNEL=lane run --max-iter $it)"; done
for it in 50 1; do echo "lane_f64_iters$it $(UMPC_QP_KERNEL=lane run --max-iter $it --dtype f64)"; done
