set -o pipefail
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1; }
echo "config3_default_k500"; run
echo "config3_k20"; run --steps 20 --warmup 5
echo "config3_euler"; run --plant euler
echo "config3_qp_only"; run --nsub 0
echo "config3_cpp_kernel"; UMPC_NO_ASM_STEP=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1
echo "config2_f64"; run --dtype f64 --batch 4096 --plant euler --steps 20 --warmup 5
echo "config5_mc"; run --monte-carlo --batch 131072 --steps 100 --warmup 100
echo "config4_p5f"; python bench.py --workload p5f --steps 20 --warmup 5 2>/dev/null | tail -1
echo "dropin"; python tools/time_dropin.py
