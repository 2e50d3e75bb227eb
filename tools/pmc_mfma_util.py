"""Aggregate a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE counter_collection.csv per kernel.
usage: pmc_mfma_util.py <counter_collection.csv> <out.json> <description> [commit]"""
import csv, json, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); ndisp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); ndisp[k].add(r["Dispatch_Id"])
out = {"what": sys.argv[3], "commit": sys.argv[4] if len(sys.argv) > 4 else None,
       "note": "mfma_util = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs); GRBM_GUI_ACTIVE is reported summed over the 8 XCDs; counter collection serialises dispatches, so durations here are not the overlapped production timeline",
       "kernels": {}}
for k, c in agg.items():
    g = c.get("GRBM_GUI_ACTIVE", 0.0); m = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    out["kernels"][k] = {"dispatches": len(ndisp[k]), "SQ_VALU_MFMA_BUSY_CYCLES": m, "GRBM_GUI_ACTIVE_sum": g,
                         "mfma_util_pct_of_1024_simds": (100.0 * m / (g / 8.0 * 1024.0)) if g else 0.0}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE_sum"])[:8]: print(k, v)
