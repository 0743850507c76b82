# headline fp32 kernel: ADMM iterations vs the rest (K = 500 launches)
set -o pipefail
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "iters50 $(run)"
echo "iters25 $(run --max-iter 25)"
echo "iters1 $(run --max-iter 1)"
echo "iters50_noplant $(run --nsub 0)"
echo "iters1_noplant $(run --nsub 0 --max-iter 1)"
