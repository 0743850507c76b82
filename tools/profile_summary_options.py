#!/usr/bin/env python3
"""rocprofv3 CSVs of tools/run_profiles_options.sh -> profiles/rNN_pmc_config5_shard.json, rNN_pmc_f3_gain_sweep.json (HBM bytes
per robot-step of the step stream's option paths, gfx950 correction of MI355X_MICROARCH.md) and rNN_quad_B16384.json /
rNN_kernel_stats_quad_B16384.csv (the quad form). usage: tools/profile_summary_options.py gpurun_out/prof_opt r04"""
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")


def counters(path, kernel):
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel_Name"]:
                out.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                                                              int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return {k: [(v, d) for _, v, d in sorted(vs)] for k, vs in out.items()}


K = 100
for cfg, name, B, alg, what in (("c5", "pmc_config5_shard", 131072, 1208 + 16, "--monte-carlo --batch 131072"),
                                ("f3", "pmc_f3_gain_sweep", 65536, 1208 + 32, "--gain-sweep"),
                                ("q", "pmc_quad_B16384", 16384, 1208, "--batch 16384")):
    kern = "umpc_rollout_asm_quad_kernel" if cfg == "q" else "umpc_rollout_asm_kernel"
    fkb = counters(os.path.join(src, cfg + "_fetch", "pmc_counter_collection.csv"), kern)["FETCH_SIZE"][-1][0]
    wkb = counters(os.path.join(src, cfg + "_write", "pmc_counter_collection.csv"), kern)["WRITE_SIZE"][-1][0]
    per = dict(read_corrected=2 * fkb * 1024 / (B * K), written=wkb * 1024 / (B * K), algorithmic=alg)
    j = dict(command="rocprofv3 --output-format csv --pmc FETCH_SIZE -- python3 bench.py --no-cpu-baseline --no-side-configs "
                     "--no-precondition --steps 100 --warmup 100 %s ; the same with --pmc WRITE_SIZE (separate passes, no trace "
                     "domains; timed dispatch = the second 100-step launch); tools/run_profiles_options.sh" % what,
             kernel=kern, batch=B, dtype="f32", plant="rk4", steps_per_launch=K, max_iter=50, nsub=25,
             FETCH_SIZE_KB_per_launch=fkb, WRITE_SIZE_KB_per_launch=wkb,
             correction="MI355X_MICROARCH.md HBM section: gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads -> read bytes = "
                        "2 x FETCH_SIZE x 1024 (upper bound for 4 B / lane accesses); WRITE_SIZE x 1024 exact",
             per_unit_bytes=per, ratio_to_algorithmic=(per["read_corrected"] + per["written"]) / alg)
    json.dump(j, open(os.path.join(dst, "%s_%s.json" % (tag, name)), "w"), indent=1)
    print(name, per, j["ratio_to_algorithmic"])
shutil.copy(os.path.join(src, "ktq", "kt_kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats_quad_B16384.csv" % tag))
kern, B = "umpc_rollout_asm_quad_kernel", 16384
c1 = counters(os.path.join(src, "q_sq1", "pmc_counter_collection.csv"), kern)
c2 = counters(os.path.join(src, "q_sq2", "pmc_counter_collection.csv"), kern)
per = {k: v[-1][0] for k, v in list(c1.items()) + list(c2.items())}
waves = per["SQ_WAVES"]
dur_ns = c2["SQ_WAVE_CYCLES"][-1][1]
sq = dict(kernel=kern, batch=B, robots_per_wave=16, steps_per_launch=K, waves=waves, per_launch=per,
          per_wave_step=dict(valu_instructions=per["SQ_INSTS_VALU"] / waves / K, salu=per["SQ_INSTS_SALU"] / waves / K,
                             lds=per["SQ_INSTS_LDS"] / waves / K, vmem_rd=per["SQ_INSTS_VMEM_RD"] / waves / K,
                             vmem_wr=per["SQ_INSTS_VMEM_WR"] / waves / K),
          derived=dict(valu_active_fraction_of_wave_cycles=per["SQ_ACTIVE_INST_VALU"] / per["SQ_WAVE_CYCLES"],
                       wait_inst_any_fraction=per["SQ_WAIT_INST_ANY"] / per["SQ_WAVE_CYCLES"],
                       dispatch_ms_under_pmc=dur_ns / 1e6, ms_per_step_under_pmc=dur_ns / 1e6 / K))
json.dump(sq, open(os.path.join(dst, "%s_sq_counters_quad_B16384.json" % tag), "w"), indent=1)
print(json.dumps(sq["per_wave_step"]), json.dumps(sq["derived"]))
