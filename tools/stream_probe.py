"""C3 batches through the batch stream: sustained QPs/s over `nb` consecutive 4096-batches at the reference's default
settings (max_iter = 10000), depth batches in flight.  usage: stream_probe.py [nb] [depth] [count]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qpdo_amd import problems, solver

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 12
count = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
t0 = time.time()
probs = [problems.config_qp("C3", i) for i in range(count)]
print("generate %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
B = [solver.Batch(probs) for _ in range(depth)]
print("batch images %.1f s" % (time.time() - t0), flush=True)
B[0].run(verbose=0, max_iter=300)                       # warm-up
t0 = time.time(); r, _ = B[0].run(verbose=0); dt1 = time.time() - t0
print("one batch alone, default settings: %.3f s = %.0f QP/s (kernel %.3f s)" % (dt1, count / dt1, B[0].kernel_seconds), flush=True)
st = solver.BatchStream(depth=depth)
for rep in range(2):
    t0 = time.time()
    tickets = []
    done = 0
    ksum = 0.0
    for b in range(nb):
        if len(tickets) == depth:
            res, ks = st.wait(tickets.pop(0)[0]); done += len(res); ksum += ks
        tickets.append((st.submit(B[b % depth], verbose=0), b))
    t_sub = time.time() - t0
    while tickets:
        res, ks = st.wait(tickets.pop(0)[0]); done += len(res); ksum += ks
    dt = time.time() - t0
    print("stream rep %d: %d batches of %d, depth %d: %.3f s = %.0f QP/s sustained (last submit at %.3f s, kernel time summed %.2f s)" % (rep, nb, count, depth, dt, done / dt, t_sub, ksum), flush=True)
st.close()
