# config 4 (p5f): lane-per-robot specialisation with fewer robots per wave (more waves in flight) vs the wave-per-robot kernel
set -o pipefail
run() { python bench.py --workload p5f --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel'])"; }
echo "wave_kernel $(run)"
for r in 64 32 16 8 4; do echo "lane_rpw$r $(UMPC_QP_KERNEL=lane UMPC_QP_RPW=$r run)"; done
for r in 64 16 8; do echo "tables_rpw$r $(UMPC_QP_KERNEL=tables UMPC_QP_RPW=$r run)"; done
