#!/bin/bash
# B = 1 drop-in latency under the image's system ROCm runtime with runtime knobs (diagnostic): which setting explains the
# gap to a process that loaded PyTorch's bundled HIP runtime (tools/time_dropin2.py)?
run() { echo -n "$1 : "; env $1 timeout -k 10 120 python tools/time_dropin.py 2>&1 | tail -1; }
run "X=1"
run "HSA_ENABLE_INTERRUPT=0"
run "HIP_FORCE_DEV_KERNARG=1"
run "AMD_DIRECT_DISPATCH=0"
run "GPU_MAX_HW_QUEUES=1"
run "HSA_ENABLE_INTERRUPT=0 HIP_FORCE_DEV_KERNARG=1"
run "UMPC_DROPIN_TWO_LAUNCHES=1 HSA_ENABLE_INTERRUPT=0"
echo -n "torch imported first : "; timeout -k 10 120 python -c "
import torch, runpy; runpy.run_path('tools/time_dropin.py')" 2>&1 | tail -1
python - <<'PY'
import os, ctypes
import torch
print("torch hip runtime:", torch.version.hip, [l for l in open('/proc/self/maps').read().split() if 'libamdhip64' in l or 'libhsa-runtime' in l][:4])
PY
