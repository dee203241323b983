"""Per-kernel summary of the rocprofv3 --pmc passes written by scripts/gpu_pmc.sh:
   python scripts/pmc_report.py gpurun_out/pmc_pile profiles/r01/pmc_pile100k.json [profiles/pmc_solve.json]
Mean counter value per dispatch over the last ticks of the run (the settled pile).  Units and corrections
as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE
reports half the bytes of 16-byte-per-lane reads (checked here on k_integrate, a streaming kernel of known
size: 40 B read and 48 B written per body), so hbm_read = 2 * FETCH_SIZE * 1024, hbm_write = WRITE_SIZE * 1024."""
import collections
import csv
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
solve_out = sys.argv[3] if len(sys.argv) > 3 else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name in sorted(os.listdir(src)):
    f = os.path.join(src, name, "pmc_counter_collection.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["duration_ns@" + name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        agg[k]["_vgpr"] = [int(r["VGPR_Count"])]
        agg[k]["_lds"] = [int(r["LDS_Block_Size"])]
        agg[k]["_wg"] = [int(r["Workgroup_Size"])]
rep = {}
for k, v in agg.items():
    if not k.startswith("sfx::"):
        continue
    tail = lambda x: x[-max(1, len(x) // 2):]   # second half of the run
    m = {c: sum(tail(x)) / len(tail(x)) for c, x in v.items()}
    e = {"dispatches_per_pass": len(v.get("SQ_WAVES", [])), "vgprs": m["_vgpr"], "lds_bytes_per_wg": m["_lds"], "wg_size": m["_wg"]}
    if "FETCH_SIZE" in m:
        e["FETCH_SIZE_KiB"] = round(m["FETCH_SIZE"], 1)
        e["hbm_read_bytes"] = round(2 * 1024 * m["FETCH_SIZE"])
    if "WRITE_SIZE" in m:
        e["WRITE_SIZE_KiB"] = round(m["WRITE_SIZE"], 1)
        e["hbm_write_bytes"] = round(1024 * m["WRITE_SIZE"])
    if "hbm_read_bytes" in e and "hbm_write_bytes" in e:
        d = m.get("duration_ns@fetch", 0)
        e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        e["duration_us_under_pmc"] = round(d / 1e3, 2)
        e["hbm_GBps_under_pmc"] = round(e["hbm_bytes"] / d, 1) if d else None
    if "SQ_WAVES" in m:
        e["waves"] = round(m["SQ_WAVES"])
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"] or 1
        e["wave_cycles_waiting_frac"] = round(m["SQ_WAIT_ANY"] / wc, 3)
        e["wave_cycles_issuing_frac"] = round(m["SQ_ACTIVE_INST_ANY"] / wc, 3)
        e["valu_insts"] = round(m["SQ_INSTS_VALU"])
        e["lds_insts"] = round(m["SQ_INSTS_LDS"])
        e["lds_bank_conflict_frac"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 3) if m["SQ_LDS_IDX_ACTIVE"] else 0.0
    if "SQ_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # mean waves in flight while the kernel runs: wave-cycles (quad-cycle units) over the busy time
        e["grbm_gui_active"] = round(m["GRBM_GUI_ACTIVE"])
    if "SQ_WAVE_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"]:
        # GRBM_GUI_ACTIVE sums the 8 XCDs (guide, DVFS note): cycles = /8; SQ_WAVE_CYCLES counts quad-cycles
        cycles = m["GRBM_GUI_ACTIVE"] / 8.0
        e["mean_waves_in_flight"] = round(4.0 * m["SQ_WAVE_CYCLES"] / cycles, 1)
        e["occupancy_waves_per_simd"] = round(4.0 * m["SQ_WAVE_CYCLES"] / cycles / 1024.0, 2)
    if "TCC_HIT_sum" in m:
        t = m["TCC_HIT_sum"] + m["TCC_MISS_sum"]
        e["l2_hit_rate"] = round(m["TCC_HIT_sum"] / t, 3) if t else None
    rep[k] = e
json.dump(rep, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, len(rep), "kernels")
if solve_out:
    res = {}
    for cls, key in (("narrow", "k_narrow"), ("solve_pos", "k_tile_pos"), ("solve_vel", "k_tile_vel")):
        ks = [x for x in rep if key in x and "hbm_bytes" in rep[x]]
        if ks:
            k = max(ks, key=lambda x: rep[x]["dispatches_per_pass"])
            res[cls] = {"kernel": k, "hbm_bytes_per_launch": rep[k]["hbm_bytes"], "hbm_read_bytes": rep[k]["hbm_read_bytes"],
                        "hbm_write_bytes": rep[k]["hbm_write_bytes"]}
    res["source"] = out
    res["note"] = ("2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes), mean per dispatch, rocprofv3 --pmc in separate passes; "
                   "the factor 2 is the guide's gfx950 correction, checked on k_integrate in the same run")
    json.dump(res, open(solve_out, "w"), indent=1)
    print("wrote", solve_out)
