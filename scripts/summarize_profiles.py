"""Turns the raw rocprofv3 output of scripts/collect_profiles.sh into the small files kept under profiles/."""
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(out_dir, "summary")
os.makedirs(dst, exist_ok=True)
OURS = ("seed_kernel", "seed_overflow_kernel", "expand_hits_kernel", "locate_kernel", "merge_kernel", "pair_kernel", "dp_fill_kernel", "dp_score_kernel", "dp_band_kernel", "dp_band_select", "dp_prefilter_kernel", "dp_class_flag", "dp_band_classify", "dp_band_list", "dp_select_kernel", "dp_compact_kernel", "dp_scatter_kernel", "dp_trace_kernel",
        "dp_ops_fix_kernel", "build_left_tasks", "build_right_tasks", "collect_pair_hits", "gather64_kernel", "prep_scan_kernel", "mark_done", "failed_flag",
        "list_failed", "single_cand", "single_task", "mate_rescue_task", "occ_count_kernel", "occ_build_kernel", "text_build_kernel", "lkt_hist_kernel")

def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


stats = newest(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"))
if stats:
    rows = list(csv.reader(open(stats[0])))
    keep = [rows[0]] + [r for r in rows[1:] if any(k in r[0] for k in OURS)]
    with open(os.path.join(dst, "%s_kernel_stats_bench_1g.csv" % tag), "w", newline="") as f:
        csv.writer(f).writerows(keep)
line = [l for l in open(os.path.join(out_dir, "bench_under_rocprof.json")) if l.startswith("{")]
if line:
    open(os.path.join(dst, "%s_bench_under_rocprof.json" % tag), "w").write(line[-1])


def counters(sub, name):
    per = {}
    for path in newest(os.path.join(out_dir, sub, "**", "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != name:
                continue
            k = r["Kernel_Name"].split("(")[0]
            per.setdefault(k, []).append(float(r["Counter_Value"]))
    return per


seed_stage = ("seed_kernel", "seed_overflow_kernel", "expand_hits_kernel", "locate_kernel")
fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
bench = json.loads([l for l in open(os.path.join(out_dir, "bench_fetch.json")) if l.startswith("{")][-1])
per_launch = {}
for k in sorted(set(fetch) | set(write)):
    per_launch[k] = {}
    if k in fetch:
        per_launch[k]["FETCH_SIZE_KiB_per_launch"] = fetch[k][-1] if k in seed_stage else sum(fetch[k]) / len(fetch[k])
        per_launch[k]["launches"] = len(fetch[k])
    if k in write:
        per_launch[k]["WRITE_SIZE_KiB_per_launch"] = write[k][-1] if k in seed_stage else sum(write[k]) / len(write[k])
# one bench step launches each seed-stage kernel once; with the 17-mer pre-filter resident the run has an untimed step
# before it that ignores the filter (reference event counts): the LAST launch of each kernel is the timed configuration,
# the first the reference walk
traffic = 1024.0 * sum((fetch.get(k, [0])[-1]) + (write.get(k, [0])[-1]) for k in seed_stage)
traffic_ref = 1024.0 * sum((fetch.get(k, [0])[0]) + (write.get(k, [0])[0]) for k in seed_stage)
g = per_launch.get("gather64_kernel", {})
doc = {
    "source": "scripts/collect_profiles.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-include-regex '<this repo kernels>' -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline",
    "workload": {"genome_mbp": bench["config"]["genome_mbp"], "pairs_per_step": bench["config"]["pairs_per_step_per_gpu"], "read_len": 150},
    "units": "FETCH_SIZE / WRITE_SIZE are in KiB (x1024 = bytes); gather64_kernel requests exactly %d B per launch (calibration of the unit for random 64-B sector gathers)" % (1 << 33),
    "per_kernel": per_launch,
    "gather64_calibration_ratio": (g.get("FETCH_SIZE_KiB_per_launch", 0) * 1024.0 / float(1 << 33)) if g else None,
    "seed_stage_hbm_bytes_per_launch": traffic,
    "seed_stage_hbm_bytes_per_launch_reference_walk": traffic_ref,
    "seed_stage_algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_pair"] * bench["config"]["pairs_per_step_per_gpu"],
    "seed_stage_reference_algorithm_bytes_per_launch": bench["roofline"].get("reference_algorithm", {}).get("bytes_per_launch"),
}
json.dump(doc, open(os.path.join(dst, "%s_pmc_seed_stage.json" % tag), "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("gather64_calibration_ratio", "seed_stage_hbm_bytes_per_launch", "seed_stage_algorithmic_bytes_per_launch")}))


# ---- VALU counters of the DP kernels (pass 4)
import collections
cc = newest(os.path.join(out_dir, "valu", "**", "*counter_collection.csv"))
kt = newest(os.path.join(out_dir, "valu", "**", "*kernel_trace.csv"))
if cc and kt:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    N_SIMD = 256 * 4
    per = {}
    for k, v in agg.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        ms = sum(dur[k]) / max(1, len(dur[k]))
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                      # summed over the 8 XCDs
        per[k] = {"launches": len(v["SQ_WAVE_CYCLES"]), "avg_ms": ms, "counters_per_launch": m,
                  "effective_clock_ghz": cyc / (ms * 1e6) if ms else None,
                  "valu_instructions_per_simd_cycle": m.get("SQ_INSTS_VALU", 0.0) / (cyc * N_SIMD) if cyc else None,
                  # SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md): x4 = cycles some wave of a SIMD had a VALU instruction executing
                  "valu_busy_frac": 4.0 * m.get("SQ_ACTIVE_INST_VALU", 0.0) / (cyc * N_SIMD) if cyc else None,
                  "wave_time_split": {n: m.get(c, 0.0) / m["SQ_WAVE_CYCLES"] for n, c in (("valu", "SQ_ACTIVE_INST_VALU"), ("any_instruction", "SQ_ACTIVE_INST_ANY"),
                                                                                           ("parked_on_waitcnt", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"))}}
    doc2 = {"source": "scripts/collect_profiles.sh pass 4: rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                      "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex 'dp_band_kernel|dp_fill_kernel' --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline",
            "units": "SQ_* cycle counters are quad-cycles summed over all waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs",
            "per_kernel": per}
    json.dump(doc2, open(os.path.join(dst, "%s_pmc_dp_valu.json" % tag), "w"), indent=1)
    print(json.dumps({k: {"valu_busy_frac": v["valu_busy_frac"], "clock": v["effective_clock_ghz"]} for k, v in per.items()}))
