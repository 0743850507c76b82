mkdir -p gpurun_out/r5e && python -m pytest tests/test_gpu_parity.py tests/test_r5_evidence.py -m gpu -q -k "fp64 or f64 or config2 or quad or blocks" > gpurun_out/r5e/pytest.log 2>&1; echo rc=$? >> gpurun_out/r5e/pytest.log; tail -3 gpurun_out/r5e/pytest.log
export TMPDIR=/tmp
C2="python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition --dtype f64 --batch 4096 --plant euler --steps 20 --warmup 5"
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-side-configs --dtype f64 --batch 4096 --plant euler --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 first pass %.4f loaded %.4f' % (j['ms_per_step'], (j['loaded_clocks'] or {}).get('ms_per_step', 0)))"; done
rocprofv3 --output-format csv --pmc FETCH_SIZE -d gpurun_out/r5e/c2_fetch -o pmc -- $C2 > gpurun_out/r5e/c2_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d gpurun_out/r5e/c2_write -o pmc -- $C2 > gpurun_out/r5e/c2_write.log 2>&1
python3 - <<'PY'
import csv, glob
def last(kind):
    f = glob.glob("gpurun_out/r5e/c2_%s/**/pmc_counter_collection.csv" % kind, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "umpc_rollout_kernel" in r["Kernel_Name"]]
    return v[-1]
f, w = 2 * last("fetch") * 1024 / (4096 * 20), last("write") * 1024 / (4096 * 20)
print("config 2: %.0f B read + %.0f B written per robot-step = %.2fx of 2416 B" % (f, w, (f + w) / 2416))
PY
