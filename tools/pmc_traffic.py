"""Per-launch HBM traffic of the slab SpMV from two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE) over
tools/spmv_only.py: the EpiStore slab dispatches come in three equal consecutive groups (A, A', Q).
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv, json, sys
def per_group(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "k_spmv_slab" in r["Kernel_Name"] and "EpiStore" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rows]
    g = len(vals) // 3
    return [sum(vals[i * g:(i + 1) * g]) / g for i in range(3)], g, rows[0]["Kernel_Name"] if rows else ""
f, gf, kn = per_group(sys.argv[1], "FETCH_SIZE"); w, gw, _ = per_group(sys.argv[2], "WRITE_SIZE")
alg = {"A  (CSR m x n)": 2403200004.0, "At (CSR n x m)": 2402800004.0, "Q  (CSR n x n)": 1203200004.0}
streamed = {"A  (CSR m x n)": 2e8 * 10, "At (CSR n x m)": 2e8 * 10, "Q  (CSR n x n)": 1e8 * 10}
out = {"what": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on tools/spmv_only.py C4, MI355X, round 1, kernel " + kn.split("(")[0] + " (slab-major image, 16-byte value loads, 16-bit slab-local indices: 10 B/nnz streamed); %d launches per matrix averaged" % gf,
       "unit_note": "counter values are KiB; gfx950 FETCH_SIZE counts 128-B requests at 64 B, so HBM read bytes = 2*FETCH_SIZE*1024 (env guide, HBM/rocprofv3 section)"}
for i, k in enumerate(alg):
    t = 2 * f[i] * 1024 + w[i] * 1024
    out[k] = {"FETCH_SIZE_KiB": f[i], "WRITE_SIZE_KiB": w[i], "traffic_bytes": t, "alg_bytes": alg[k], "traffic_over_alg": t / alg[k], "streamed_nnz_bytes": streamed[k]}
json.dump(out, open(sys.argv[3], "w"), indent=1); print(json.dumps(out, indent=1))
