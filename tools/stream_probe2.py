"""variants of stream_probe to find what serialises the stream: usage stream_probe2.py nb depth warm(0/1) pre_images(0/1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver
nb, depth, warm, pre = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
count = 4096
probs = [problems.config_qp("C3", i) for i in range(count)]
B0 = solver.Batch(probs)
if pre:
    B = [B0] + [solver.Batch(probs) for _ in range(depth - 1)]
B0.run(verbose=0)
if not pre:
    B = [B0] + [solver.Batch(probs) for _ in range(depth - 1)]
st = solver.BatchStream(depth=depth)
if warm:
    for t in [st.submit(b, verbose=0, max_iter=50) for b in B]:
        st.wait(t)
t0 = time.time(); tickets = []; ks_all = []; sub = []
for b in range(nb):
    if len(tickets) == depth:
        _, ks = st.wait(tickets.pop(0)); ks_all.append(ks)
    ts = time.time(); tickets.append(st.submit(B[b % depth], verbose=0)); sub.append(time.time() - ts)
while tickets:
    _, ks = st.wait(tickets.pop(0)); ks_all.append(ks)
dt = time.time() - t0
print("nb %d depth %d warm %d pre %d: %.2f s = %.0f QP/s; submit times %s; kernel s %s" % (nb, depth, warm, pre, dt, nb * count / dt, " ".join("%.2f" % s for s in sub), " ".join("%.2f" % k for k in ks_all)), flush=True)
st.close()
