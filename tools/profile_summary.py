#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs written by tools/run_profiles.sh into the committed summaries under profiles/:
   rNN_kernel_stats_k500.csv / _k20.csv  (--kernel-trace --stats), rNN_kernel_trace_rollout.csv (the step-kernel dispatches),
   rNN_pmc_traffic.json (FETCH_SIZE / WRITE_SIZE passes, gfx950 correction of MI355X_MICROARCH.md), rNN_sq_counters.json.
   usage: tools/profile_summary.py gpurun_out/r2prof r02"""
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")
KERNEL = "umpc_rollout_asm_kernel"
B = 65536


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def counters(path):
    """{counter: [value per dispatch of the step kernel, in dispatch order]}"""
    out = {}
    for r in rows(path):
        if KERNEL in r["Kernel_Name"]:
            out.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                                                          int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return {k: [(v, d) for _, v, d in sorted(vs)] for k, vs in out.items()}


for k in ("kt500", "kt20"):
    shutil.copy(os.path.join(src, k, "kt_kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, k[2:] and "k" + k[2:])))
for k, name in (("ktc2", "config2_f64"), ("ktc4", "config4_p5f")):
    if os.path.exists(os.path.join(src, k, "kt_kernel_stats.csv")):
        shutil.copy(os.path.join(src, k, "kt_kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, name)))
with open(os.path.join(dst, "%s_kernel_trace_rollout.csv" % tag), "w") as f:
    for k in ("kt500", "kt20"):
        lines = open(os.path.join(src, k, "kt_kernel_trace.csv")).read().splitlines()
        f.write("# %s\n%s\n" % (k, lines[0]))
        for ln in lines[1:]:
            if KERNEL in ln:
                f.write(ln + "\n")

K = 500
fetch = counters(os.path.join(src, "fetch", "pmc_counter_collection.csv"))["FETCH_SIZE"]
write = counters(os.path.join(src, "write", "pmc_counter_collection.csv"))["WRITE_SIZE"]
fkb, wkb = fetch[-1][0], write[-1][0]          # the timed dispatch = the second 500-step launch
alg = 1208
traffic = dict(
    command="rocprofv3 --output-format csv --pmc FETCH_SIZE -- python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition ; the same with --pmc WRITE_SIZE "
            "(separate passes, no trace domains; timed dispatch = the second 500-step launch); tools/run_profiles.sh",
    kernel=KERNEL, batch=B, dtype="f32", plant="rk4", steps_per_launch=K, max_iter=50, nsub=25,
    FETCH_SIZE_KB_per_launch=fkb, WRITE_SIZE_KB_per_launch=wkb,
    correction="MI355X_MICROARCH.md HBM section: gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads -> read bytes = "
               "2 x FETCH_SIZE x 1024 (upper bound: our reads are 4 B/lane = 256 B/wave, a width the guide does not calibrate); "
               "WRITE_SIZE x 1024 exact",
    hbm_bytes_per_launch=2 * fkb * 1024 + wkb * 1024, hbm_bytes_per_launch_uncorrected=(fkb + wkb) * 1024,
    per_robot_step_bytes=dict(read_corrected=2 * fkb * 1024 / (B * K), written=wkb * 1024 / (B * K), algorithmic=alg))
traffic["ratio_to_algorithmic"] = (traffic["per_robot_step_bytes"]["read_corrected"] + traffic["per_robot_step_bytes"]["written"]) / alg
json.dump(traffic, open(os.path.join(dst, "%s_pmc_traffic.json" % tag), "w"), indent=1)

KS = 100
c1 = counters(os.path.join(src, "sq1", "pmc_counter_collection.csv"))
c2 = counters(os.path.join(src, "sq2", "pmc_counter_collection.csv"))
per = {k: v[-1][0] for k, v in list(c1.items()) + list(c2.items())}
dur_ns = c2["SQ_WAVE_CYCLES"][-1][1]
waves = per["SQ_WAVES"]
sq = dict(
    command="rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -- "
            "python3 bench.py --no-cpu-baseline --no-side-configs --no-precondition --steps 100 --warmup 100 ; second pass --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES "
            "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE",
    kernel=KERNEL, batch=B, dtype="f32", plant="rk4", steps_per_launch=KS, max_iter=50, nsub=25, waves=waves, per_launch=per,
    per_wave_step=dict(valu_instructions=per["SQ_INSTS_VALU"] / waves / KS, salu=per["SQ_INSTS_SALU"] / waves / KS,
                       lds=per["SQ_INSTS_LDS"] / waves / KS, vmem_rd=per["SQ_INSTS_VMEM_RD"] / waves / KS,
                       vmem_wr=per["SQ_INSTS_VMEM_WR"] / waves / KS),
    derived=dict(valu_active_fraction_of_wave_cycles=per["SQ_ACTIVE_INST_VALU"] / per["SQ_WAVE_CYCLES"],
                 wait_inst_any_fraction=per["SQ_WAIT_INST_ANY"] / per["SQ_WAVE_CYCLES"],
                 wave_quad_cycles_per_wave_step=per["SQ_WAVE_CYCLES"] / waves / KS,
                 dispatch_ms_under_pmc=dur_ns / 1e6,
                 effective_clock_GHz=(per.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur_ns * 1e-9) / 1e9) if per.get("GRBM_GUI_ACTIVE") else None,
                 note="SQ cycle counters tick once per 4 clocks; GRBM_GUI_ACTIVE is summed over the 8 XCDs"))
json.dump(sq, open(os.path.join(dst, "%s_sq_counters.json" % tag), "w"), indent=1)
print(json.dumps(dict(traffic=traffic["per_robot_step_bytes"], ratio=traffic["ratio_to_algorithmic"], sq=sq["per_wave_step"],
                      derived=sq["derived"]), indent=1))
