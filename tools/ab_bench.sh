#!/bin/bash
# A/B timing of step-kernel variants (robobee3d_amd/variants/libumpc_<name>.so) on ONE box: each variant is timed
# twice, interleaved. usage: tools/ab_bench.sh base xv full
for rep in 1 2; do
  for v in "$@"; do
    UMPC_LIB=$PWD/robobee3d_amd/variants/libumpc_$v.so timeout -k 10 200 python bench.py --steps 500 --warmup 500 --no-cpu-baseline 2>/dev/null | \
      python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep$rep K=500 kernel ms/step %.5f' % (j['roofline']['kernel_ms']/j['roofline']['steps_per_launch']))"
  done
done
