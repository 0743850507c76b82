# fp64 config 2: cost of the Ruiz passes (variant built with UMPC_SCALING_ITERS=1: timing only, different numbers)
set -o pipefail
run() { python bench.py --dtype f64 --batch 4096 --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "ruiz10_iters1_noplant $(run --nsub 0 --max-iter 1)"
echo "ruiz1_iters1_noplant $(UMPC_LIB=robobee3d_amd/variants/libumpc_ruiz1.so run --nsub 0 --max-iter 1)"
echo "ruiz10_iters50_euler $(run --plant euler)"
