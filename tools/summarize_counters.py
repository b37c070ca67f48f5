#!/usr/bin/env python3
"""Condense the SQ passes of tools/collect_profiles.sh:
   <out>/<tag>_sq_counters.json   per kernel: every counter summed over its launches + launch count
   <out>/<tag>_lds_roofline.json  the `roofline_lds` object bench.py embeds (dominant kernel: audio Rips), every
                                  figure recomputable from <tag>_sq_counters.json alone:
       lds_bytes_per_clk_per_cu   SQ_LDS_IDX_ACTIVE (LDS-array cycles, all CUs) -> share of the CU-cycles of the
                                  kernel in which the LDS array was busy, x 128 B/clk (the ds_read_b32 / b64 width
                                  that dominates; MI355X_MICROARCH.md LDS table) = achieved B/clk/CU
       valu_issue_util            SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x waves per SIMD ... reported as the share of
                                  SIMD-cycles that issued a vector instruction
       wait_frac                  SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave parked at s_waitcnt / barrier)
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def short(name):
    name = re.sub(r"\(.*$", "", name).strip()
    return name.replace("void ", "")


acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(lambda: defaultdict(int))
for d in sorted(glob.glob(os.path.join(out, "pmc_sq*"))):
    if not os.path.isdir(d):
        continue
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]] += 1
res = {}
for k, c in acc.items():
    if not any(s in k for s in ("rips_", "corr_dist", "wasserstein", "eeg_", "features", "tau_kernel", "recording_rows",
                                "h1_order", "diagram_finish")):
        continue
    res[k] = {"launches": max(launches[k].values()), **{n: v for n, v in sorted(c.items())}}
json.dump(res, open(os.path.join(out, f"{tag}_sq_counters.json"), "w"), indent=1)


def roof(kname_part):
    for k, c in res.items():
        if kname_part in k:
            g = lambda n: c.get(n)
            wc, busy_cu = g("SQ_WAVE_CYCLES"), g("SQ_BUSY_CU_CYCLES")
            o = {"kernel": k, "launches": c["launches"]}
            if wc:
                for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU",
                          "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS"):
                    if g(n) is not None:
                        o[n.lower().replace("sq_", "") + "_per_wave_cycle"] = round(g(n) / wc, 4)
            if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None:
                o["lds_bank_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
            if g("SQ_LDS_IDX_ACTIVE") and busy_cu:
                # SQ_BUSY_CU_CYCLES: cycles a CU had waves, summed over CUs (quad-cycle units like the other SQ cycle
                # counters, MI355X_MICROARCH.md cycle-constants table); LDS_IDX_ACTIVE in the same units
                share = g("SQ_LDS_IDX_ACTIVE") / busy_cu
                o["lds_array_busy_share_of_cu_cycles"] = round(share, 4)
                o["lds_bytes_per_clk_per_cu_achieved"] = round(128.0 * share, 2)
                o["lds_bytes_per_clk_per_cu_peak"] = 128.0
                o["lds_frac_of_peak"] = round(share, 4)
            for n in ("SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_WAVES"):
                if g(n) is not None:
                    o[n.lower().replace("sq_", "")] = g(n)
            return o
    return None


lds = {"bound": "lds/issue", "unit": "B/clk/CU", "peak": 128.0,
       "source": f"profiles/{tag}_sq_counters.json (rocprofv3 --pmc, tools/collect_profiles.sh)",
       "rips_cloud": roof("rips_cloud_kernel<512, 1, unsigned int, false"), "rips_dm": roof("rips_dm_kernel<256, 1, 1"),
       "eeg_fused": roof("eeg_window_kernel<3, false, 1, false")}
# vector wave-instructions per window pair of the step, kernel by kernel (the audio kernel runs 8 waves per window:
# its wave count gives the windows these passes processed)
cloud = next((c for k, c in res.items() if "rips_cloud_kernel<512, 1, unsigned int, false" in k), None)
if cloud and cloud.get("SQ_WAVES") and cloud.get("SQ_INSTS_VALU"):
    n_win = cloud["SQ_WAVES"] / 8.0
    per = {k: round(c["SQ_INSTS_VALU"] / n_win, 1) for k, c in res.items() if c.get("SQ_INSTS_VALU")}
    lds["valu_per_window"] = {"windows_profiled": int(n_win), "by_kernel": per, "total": round(sum(per.values()), 1),
                              "note": "SQ_INSTS_VALU (wave-instructions) of every kernel of the step divided by the window "
                                      "pairs processed; bench.py turns it into `roofline_valu` with its own pass time"}
json.dump(lds, open(os.path.join(out, f"{tag}_lds_roofline.json"), "w"), indent=1)
print(json.dumps(lds, indent=1))
