"""Average rocprofv3 --pmc counters (and kernel durations) per launch of the kernels whose name contains <substr>.
    python tools/pmc_summary.py <substr> <out_dir> [<out_dir> ...]      (each dir = one `rocprofv3 --pmc ... --kernel-trace -d <dir>` pass)"""
import collections, csv, glob, json, sys

sub, dirs = sys.argv[1], sys.argv[2:]
out = {"kernel_substring": sub, "counters_avg_per_launch": {}, "kernel_us": {}}
durs = []
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc, n = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
        for k in acc:
            out["counters_avg_per_launch"][k] = acc[k] / n[k]
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if durs:
    out["kernel_us"] = {"mean": sum(durs) / len(durs), "min": min(durs), "max": max(durs), "launches": len(durs)}
c = out["counters_avg_per_launch"]
der = {}
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c:
    # MFMA_BUSY is summed over SIMDs (4 per CU), BUSY_CU over CUs
    der["mfma_busy_fraction_of_cu_busy_cycles"] = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 4) / c["SQ_BUSY_CU_CYCLES"] if c["SQ_BUSY_CU_CYCLES"] else None
    if durs:
        der["implied_clock_ghz"] = c["SQ_BUSY_CU_CYCLES"] / 256 / (out["kernel_us"]["mean"] * 1e3)
if "GRBM_GUI_ACTIVE" in c and durs:
    der["clock_ghz_from_grbm"] = c["GRBM_GUI_ACTIVE"] / 8 / (out["kernel_us"]["mean"] * 1e3)
if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
    der["lds_bank_conflict_fraction"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
    if k in c and c.get("SQ_WAVE_CYCLES"):
        der[k + "_fraction_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
out["derived"] = der
print(json.dumps(out, indent=1))
