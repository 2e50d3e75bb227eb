import os, sys
sys.path.insert(0, "/root/repo")
os.environ["QPDO_SMALL_PROF"] = "1"
from qpdo_amd import problems, solver
probs = [problems.config_qp("C3", i) for i in range(512)]
B = solver.Batch(probs)
B.run(verbose=0)
B.run(verbose=0)
