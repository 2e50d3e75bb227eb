"""Per-kernel HBM traffic of a profiled bench.py run from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
passes, --output-format csv).  Averages over the REAL launches of each kernel (latched no-op launches of a converged inner
solve move < 1 MiB and are excluded).  gfx950: HBM read bytes = 2 * FETCH_SIZE KiB * 1024 (MI355X_MICROARCH.md, HBM section).
usage: pmc_inner.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <description> [commit] [bench.json]
commit and the bench line of an unprofiled run of the same command record WHEN and ON WHAT LAUNCH GEOMETRY the counters were taken
(alg_bytes_per_launch_avg of the dominant kernel): bench.py prints roofline.traffic only while the live kernel still matches."""
import collections, csv, json, sys


def load(path, counter):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        short = k.split("(")[0].replace("void ", "")
        if "<" in short:
            short = short.split("<")[0] + "<" + short.split("<")[1].split(",")[0].rstrip(">") + ">"
        per[short].append(float(r["Counter_Value"]))
    return per


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"what": sys.argv[4],
       "unit_note": "counter values are KiB; gfx950 FETCH_SIZE counts 128-B requests at 64 B, so HBM read bytes = 2*FETCH_SIZE*1024 (env guide, HBM/rocprofv3 section)"}
tot = {k: sum(v) for k, v in f.items()}
for k in sorted(tot, key=lambda k: -tot[k])[:12]:
    fv = f[k]; wv = w.get(k, [])
    real = [i for i, v in enumerate(fv) if v >= 1024.0]           # >= 1 MiB fetched: a launch that did its work
    if not real:
        continue
    fa = sum(fv[i] for i in real) / len(real)
    wa = (sum(wv[i] for i in real if i < len(wv)) / len(real)) if wv else 0.0
    out[k] = {"launches": len(fv), "real_launches": len(real), "FETCH_SIZE_KiB_avg": fa, "WRITE_SIZE_KiB_avg": wa,
              "traffic_bytes_avg": 2 * fa * 1024 + wa * 1024}
if len(sys.argv) > 5:
    out["commit"] = sys.argv[5]
if len(sys.argv) > 6:
    line = [l for l in open(sys.argv[6]) if l.startswith("{")][-1]
    roof = json.loads(line)["roofline"]
    if "k_spmv_slab<EpiSchurW>" in out and "EpiSchurW" in roof.get("kernel", ""):
        out["k_spmv_slab<EpiSchurW>"]["alg_bytes_per_launch_avg"] = roof["alg_bytes_per_launch"]
        out["k_spmv_slab<EpiSchurW>"]["live_avg_launch_s"] = roof["avg_launch_s"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
