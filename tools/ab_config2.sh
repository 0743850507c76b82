#!/bin/bash
# A/B of two builds of the library on BASELINE configs[1] (B = 4 096, fp64), one box, three interleaved repetitions.
# usage: tools/ab_config2.sh <variant> <variant> ...   (robobee3d_amd/variants/libumpc_<variant>.so)
for rep in 1 2 3; do
  for v in "$@"; do
    UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_$v.so python bench.py --no-cpu-baseline --no-side-configs --dtype f64 --batch 4096 --plant euler --steps 200 --warmup 50 2>/dev/null | \
      python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep$rep K=200 ms/step %.5f  (kernel %s)' % (j['ms_per_step'], j['roofline']['kernel']))"
    UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_$v.so python bench.py --no-cpu-baseline --no-side-configs --dtype f64 --batch 4096 --plant euler --steps 20 --warmup 5 2>/dev/null | \
      python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep$rep K=20 W=5 first pass %.5f  loaded %.5f' % (j['ms_per_step'], (j['loaded_clocks'] or {}).get('ms_per_step', 0)))"
  done
done
