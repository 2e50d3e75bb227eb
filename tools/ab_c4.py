"""A/B timing of two builds of libqpdo_amd.so on one box: the C4 cold-start solve, alternating processes.
usage: ab_c4.py libA.so libB.so [reps]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r"""
import sys, time, json
sys.path.insert(0, %r)
from qpdo_amd import problems, solver
p = problems.config_qp("C4")
s = solver.QPDO().setup(p["Q"], p["q"], p["A"], p["l"], p["u"], Qstype=-1, verbose=0)
out = []
for _ in range(2):
    t0 = time.time(); r = s.solve(); solver.lib().qpdo_amd_sync(s._w); dt = time.time() - t0
    out.append(dict(t=dt, it=r["info"]["iterations"], st=r["info"]["status_val"], cg=s.stats()["lin_iters"]))
print(json.dumps(out))
""" % ROOT
libs = sys.argv[1:3]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for rep in range(reps):
    for lib in libs:
        env = dict(os.environ, QPDO_AMD_LIB=os.path.abspath(lib))
        o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("[")]
        print(os.path.basename(lib), line[-1] if line else o.stderr[-500:], flush=True)
