"""MFMA utilisation of k_ldl_syrk split into the wide launches (trailing updates over a whole outer panel: 1-D grids of the XCD-aware order,
Grid_Size_Y == 1 and more than 2048 workgroups... identified here by a grid above `min_wgs` workgroups) and the rest (updates inside an
outer panel).  usage: pmc_mfma_wide.py <counter_collection.csv> <out.json> [min_wgs]"""
import csv, json, sys, collections
min_wgs = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
d = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_ldl_syrk" not in r["Kernel_Name"]:
        continue
    key = r["Dispatch_Id"]
    d[key][r["Counter_Name"]] = d[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    d[key]["wgs"] = int(r["Grid_Size"]) // 256
agg = {"wide": [0.0, 0.0, 0], "narrow": [0.0, 0.0, 0]}
for k, c in d.items():
    cls = "wide" if c["wgs"] >= min_wgs else "narrow"
    agg[cls][0] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); agg[cls][1] += c.get("GRBM_GUI_ACTIVE", 0.0); agg[cls][2] += 1
out = {"what": "k_ldl_syrk, rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over tools/dense_lab.py 10000 (counter collection serialises dispatches)",
       "note": "util = MFMA busy cycles / (GUI_ACTIVE / 8 XCDs x 1024 SIMDs); wide = launches of at least %d workgroups (updates by a whole outer panel), narrow = the updates inside an outer panel and the small late launches" % min_wgs}
for cls, (m, g, n) in agg.items():
    out[cls] = {"dispatches": n, "mfma_busy_cycles": m, "gui_active_sum": g, "mfma_util_pct_of_1024_simds": (100.0 * m / (g / 8.0 * 1024.0)) if g else 0.0}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
