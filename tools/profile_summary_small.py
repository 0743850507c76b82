#!/usr/bin/env python3
"""Summaries of tools/run_profiles_small.sh (counter passes of config 4 = planar p5f fp32 at B = 16 384 and config 2 =
uprightmpc2 fp64 at B = 4 096) -> profiles/rNN_pmc_config4_p5f.json, rNN_pmc_config2_f64.json. Same conventions as
profile_summary.py (gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md; SQ cycle counters tick once per 4 clocks).
usage: tools/profile_summary_small.py gpurun_out/r2small r02"""
import csv
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")
CFG = {
    "c4": dict(name="config4_p5f", kernel="bqp_fixed_p5f10_asm_kernel", B=16384, dtype="f32", units_per_dispatch=1,
               unit="robot-tick", alg_bytes=(2 * (87 + 2 * 164) + 15) * 4,
               command="bench.py --no-cpu-baseline --no-side-configs --no-precondition --workload p5f --steps 20 --warmup 5"),
    "c2": dict(name="config2_f64", kernel="umpc_rollout_kernel", B=4096, dtype="f64", units_per_dispatch=20,
               unit="robot-step", alg_bytes=2 * 1208,
               command="bench.py --no-cpu-baseline --no-side-configs --no-precondition --dtype f64 --batch 4096 --plant euler --steps 20 --warmup 5"),
}


def counters(path, kernel):
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel_Name"]:
                out.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                                                              int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return {k: [(v, d) for _, v, d in sorted(vs)] for k, vs in out.items()}


for cfg, c in CFG.items():
    get = lambda sub: counters(os.path.join(src, "%s_%s" % (cfg, sub), "pmc_counter_collection.csv"), c["kernel"])
    fetch, write = get("fetch")["FETCH_SIZE"], get("write")["WRITE_SIZE"]
    c1, c2 = get("sq1"), get("sq2")
    # the timed dispatches are the last `steps` ones (config 4: one per tick) / the last one (config 2: one 20-step launch)
    n = 20 if c["units_per_dispatch"] == 1 else 1
    mean = lambda xs: sum(v for v, _ in xs[-n:]) / n
    fkb, wkb = mean(fetch), mean(write)
    per = {k: mean(v) for k, v in list(c1.items()) + list(c2.items())}
    units = c["B"] * c["units_per_dispatch"]
    waves = per["SQ_WAVES"]
    dur_ns = sum(d for _, d in c2["SQ_WAVE_CYCLES"][-n:]) / n
    out = dict(
        command="rocprofv3 --output-format csv --pmc <counters> -- python3 " + c["command"] +
                " ; separate passes: FETCH_SIZE | WRITE_SIZE | SQ_INSTS_* SQ_WAVES | SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY "
                "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE (tools/run_profiles_small.sh); mean over the timed dispatches",
        kernel=c["kernel"], batch=c["B"], dtype=c["dtype"], unit=c["unit"], units_per_dispatch=units, waves=waves,
        FETCH_SIZE_KB_per_dispatch=fkb, WRITE_SIZE_KB_per_dispatch=wkb,
        correction="read bytes = 2 x FETCH_SIZE x 1024 (gfx950, upper bound for 256-B wave reads), WRITE_SIZE x 1024 exact",
        per_unit_bytes=dict(read_corrected=2 * fkb * 1024 / units, written=wkb * 1024 / units, algorithmic=c["alg_bytes"]),
        per_wave_unit=dict(valu_instructions=per["SQ_INSTS_VALU"] / waves / c["units_per_dispatch"],
                           salu=per["SQ_INSTS_SALU"] / waves / c["units_per_dispatch"],
                           lds=per["SQ_INSTS_LDS"] / waves / c["units_per_dispatch"],
                           vmem_rd=per["SQ_INSTS_VMEM_RD"] / waves / c["units_per_dispatch"],
                           vmem_wr=per["SQ_INSTS_VMEM_WR"] / waves / c["units_per_dispatch"]),
        derived=dict(valu_active_fraction_of_wave_cycles=per["SQ_ACTIVE_INST_VALU"] / per["SQ_WAVE_CYCLES"],
                     wait_inst_any_fraction=per["SQ_WAIT_INST_ANY"] / per["SQ_WAVE_CYCLES"],
                     dispatch_ms_under_pmc=dur_ns / 1e6,
                     effective_clock_GHz=per["GRBM_GUI_ACTIVE"] / 8 / (dur_ns * 1e-9) / 1e9))
    out["ratio_to_algorithmic"] = (out["per_unit_bytes"]["read_corrected"] + out["per_unit_bytes"]["written"]) / c["alg_bytes"]
    json.dump(out, open(os.path.join(dst, "%s_pmc_%s.json" % (tag, c["name"])), "w"), indent=1)
    print(c["name"], json.dumps(dict(bytes=out["per_unit_bytes"], ratio=out["ratio_to_algorithmic"], per_wave=out["per_wave_unit"],
                                     derived=out["derived"]), indent=1))
