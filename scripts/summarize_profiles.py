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
OURS = ("reduce_", "floor_", "cand_first", "dp_len_class", "dp_nonempty", "dp_gather_tasks", "dp_scatter_list", "census", "dp_thermo_kernel", "thermo_flag", "single_hit", "rescue_hit", "seed_kernel", "seed_overflow_kernel", "expand_hits_kernel", "locate_kernel", "merge_kernel", "pair_kernel", "dp_fill_kernel", "dp_score_kernel", "dp_band_kernel", "dp_band_select", "dp_prefilter_kernel", "dp_class_flag", "dp_band_classify", "dp_band_list", "dp_select_kernel", "dp_compact_kernel", "dp_scatter_kernel", "dp_trace_kernel",
        "dp_ops_fix_kernel", "build_left_tasks", "build_right_tasks", "collect_pair_hits", "gather64_kernel", "prep_scan_kernel", "mark_done", "failed_flag",
        "list_failed", "single_cand", "single_task", "mate_rescue_task", "occ_count_kernel", "occ_build_kernel", "text_build_kernel", "lkt_hist_kernel")

def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


stats = newest(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"))
if stats:
    rows = list(csv.reader(open(stats[0])))
    keep = [rows[0]] + [r for r in rows[1:] if any(k in r[0] for k in OURS)]
    with open(os.path.join(dst, "%s_kernel_stats_bench.csv" % tag), "w", newline="") as f:
        csv.writer(f).writerows(keep)
    # every kernel of the run (library kernels of hipcub and torch included), the 80 largest by total time: what the filtered table leaves out
    with open(os.path.join(dst, "%s_kernel_stats_bench_all_top80.csv" % tag), "w", newline="") as f:
        csv.writer(f).writerows(rows[:81])
# the same trace restricted to the steps: every kernel (copy and fill kernels of the runtime included) that starts after the first seed_kernel does,
# i.e. without the building of the replica -- calls, total and share of the window
trace = newest(os.path.join(out_dir, "trace", "**", "*kernel_trace.csv"))
if trace:
    import collections as _c
    recs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(trace[0]))]
    seeds = [a for a, b, k in recs if k.startswith("seed_kernel")]
    if seeds:
        t0 = min(seeds)
        t1 = max(b for a, b, k in recs)
        agg = _c.defaultdict(lambda: [0, 0])
        for a, b, k in recs:
            if a >= t0:
                name = k.split("(")[0][:110]
                agg[name][0] += 1; agg[name][1] += b - a
        busy = sum(v[1] for v in agg.values())
        with open(os.path.join(dst, "%s_kernel_stats_steps_only.csv" % tag), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "ShareOfKernelTime", "window_ns=%d" % (t1 - t0), "kernel_time_ns=%d" % busy])
            for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
                w.writerow([name, n, tot, tot // n, "%.4f" % (tot / busy)])
line = [l for l in open(os.path.join(out_dir, "bench_under_rocprof.json")) if l.startswith("{")]
if line:
    open(os.path.join(dst, "%s_bench_under_rocprof.json" % tag), "w").write(line[-1])


# ---- VALU counters of the DP kernels (pass 4)
import collections
for sub, suffix, cmd in (("valu", "", "python3 bench.py --pmc-child nt (one single-context step of the headline batch through 3 chunks)"),
                         ("valu_human", "_human_filter", "python3 bench.py --pmc-child human (one step of the human-filter leg: ONE 2.95 Gbp index, every optional table, soap4.ini)")):
  cc = newest(os.path.join(out_dir, sub, "**", "*counter_collection.csv"))
  kt = newest(os.path.join(out_dir, sub, "**", "*kernel_trace.csv"))
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
      doc2 = {"source": "scripts/collect_profiles.sh: rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                        "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex '<DP and seed-stage kernels>' --kernel-trace -- " + cmd,
              "units": "SQ_* cycle counters are quad-cycles summed over all waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs",
              "per_kernel": per}
      json.dump(doc2, open(os.path.join(dst, "%s_pmc_sq_counters%s.json" % (tag, suffix)), "w"), indent=1)
      print(json.dumps({k: {"valu_busy_frac": v["valu_busy_frac"], "clock": v["effective_clock_ghz"]} for k, v in per.items()}))
