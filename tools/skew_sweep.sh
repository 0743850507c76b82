#!/bin/bash
# Start-skew sweep of the all-assembly step kernel on ONE box (K = 20 driver shape and K = 500), kernel ms per step.
for rep in 1 2; do
  for g in 4 8; do
    for us in 0 2 5 10 20 40; do
      for shape in "20 5" "500 500"; do
        set -- $shape
        UMPC_ASM_SKEW_US=$us UMPC_ASM_SKEW_GROUPS=$g timeout -k 10 200 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-side-configs 2>/dev/null | \
          python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep$rep groups=$g skew_us=$us K=$1 ms/step wall %.5f kernel %.5f' % (j['ms_per_step'], j['roofline']['kernel_ms']/j['roofline']['steps_per_launch']))"
      done
    done
  done
done
