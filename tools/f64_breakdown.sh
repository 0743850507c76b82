# where the fp64 step (config 2) spends its time: ADMM iterations vs the rest, plant vs none
set -o pipefail
run() { python bench.py --dtype f64 --batch 4096 --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "iters50_euler $(run --plant euler)"
echo "iters1_euler $(run --plant euler --max-iter 1)"
echo "iters50_noplant $(run --nsub 0)"
echo "iters1_noplant $(run --nsub 0 --max-iter 1)"
echo "iters25_noplant $(run --nsub 0 --max-iter 25)"
echo "f32_b4096_iters50 $(python bench.py --batch 4096 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
echo "f32_b4096_cpp $(UMPC_NO_ASM_STEP=1 python bench.py --batch 4096 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
