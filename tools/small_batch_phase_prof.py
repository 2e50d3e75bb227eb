"""In-kernel phase times of the slowest item of a fused batch (QPDO_SMALL_PROF=1: 100 MHz ticks accumulated by lane 0; the four
"ls:" entries are sub-phases of the linesearch).  usage: small_batch_phase_prof.py [max_iter]   (default: the reference's 10000 --
the slowest item is then one that stalls; a small max_iter, e.g. 40, shows an ordinary solve instead)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QPDO_SMALL_PROF"] = "1"
from qpdo_amd import problems, solver
probs = [problems.config_qp("C3", i) for i in range(512)]
B = solver.Batch(probs)
kw = dict(max_iter=int(sys.argv[1])) if len(sys.argv) > 1 else {}
B.run(verbose=0, **kw)
B.run(verbose=0, **kw)
