# fp64 config 2: A/B of assembly-loop fetch parameters (variants built with UMPC_ASM64_AHEAD / UMPC_ASM64_MERGE), same box
set -o pipefail
run() { python bench.py --dtype f64 --batch 4096 --steps 20 --warmup 5 --plant euler --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
echo "default(14,1) $(run)"
for v in a20m1 a14m2 a20m3 a28m3; do echo "$v $(UMPC_LIB=robobee3d_amd/variants/libumpc_$v.so run)"; done
done
