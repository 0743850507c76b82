#!/bin/bash
# A/B timing of fp64 step-kernel variants (robobee3d_amd/variants/libumpc_<name>.so) on ONE box, config 2
# (fp64, B = 4096, Euler plant): each variant twice, interleaved. usage: tools/ab_bench64.sh s0 s3 s4
for rep in 1 2; do
  for v in "$@"; do
    UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_$v.so timeout -k 10 200 python bench.py --dtype f64 --batch 4096 --plant euler --steps 100 --warmup 20 --no-cpu-baseline --no-side-configs 2>/dev/null | \
      python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep$rep fp64 B=4096 ms/step %.5f  kernel %.5f' % (j['ms_per_step'], j['roofline']['kernel_ms']/j['roofline']['steps_per_launch']))"
  done
done
