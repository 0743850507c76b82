#!/bin/bash
# A/B timing of step-kernel variants (robobee3d_amd/variants/libumpc_<name>.so) on ONE box, three interleaved repetitions, at
# K = 500 per launch (loaded clock) and in the driver's shape (--steps 20 --warmup 5: first pass and loaded pass).
# usage: tools/ab_bench2.sh <variant> <variant> ...
for rep in 1 2 3; do
  for v in "$@"; do
    UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_$v.so timeout -k 10 200 python bench.py --steps 500 --warmup 500 --no-cpu-baseline --no-side-configs 2>/dev/null | \
      python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep$rep K=500 kernel ms/step %.5f' % (j['roofline']['kernel_ms']/j['roofline']['steps_per_launch']))"
    UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_$v.so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs 2>/dev/null | \
      python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep$rep K=20 W=5 first pass ms/step %.5f   loaded %.5f' % (j['ms_per_step'], (j['loaded_clocks'] or {}).get('ms_per_step', float('nan'))))"
  done
done
