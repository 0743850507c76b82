# fp64 at batches beyond one wave per CU: the LDS-resident / assembly-loop kernel (one workgroup per CU, in rounds) against the
# all-C++ kernel (four waves per CU)
set -o pipefail
run() { python bench.py --dtype f64 --steps 10 --warmup 2 --plant euler --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for B in 16384 32768 65536; do
echo "B=$B cpp_kernel $(UMPC_NO_F64_LDS=1 run --batch $B)"
echo "B=$B asm64_rounds $(UMPC_F64_LDS_MAX_GRID=100000 run --batch $B)"
done
