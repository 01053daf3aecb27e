cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_misc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "dp_prefilter_kernel|dp_band_classify|collect_pair_hits|dp_trace_kernel|single_cand_kernel|prep_scan" --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-child --steps 1 --warmup 0 > $OUT/bench.json 2> $OUT/err.txt
echo rc=$?
python3 - <<PY
import csv, glob, collections
cc = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
kt = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc[0])):
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt[0])):
    dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in agg.items():
    m = {c: x[-1] for c, x in v.items()}
    ms = dur[k][-1]
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k[:50], "ms %.3f" % ms, "valu_busy %.2f" % (4 * m["SQ_ACTIVE_INST_VALU"] / (cyc * 1024)), "wave split: valu %.2f any %.2f wait_any %.2f wait_inst %.2f" % (
              m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
