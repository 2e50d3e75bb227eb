"""The oracle reproduces the committed golden vectors (regression pin of the checker itself)."""
import numpy as np
import pytest

from helpers import close_vec, golden_problem, load_golden
from oracle import binding as ob

GOLD = load_golden()


@pytest.mark.parametrize("name", sorted(GOLD))
def test_oracle_matches_golden(name):
    g = GOLD[name]
    p = golden_problem(g["spec"])
    o = ob.OracleSolver(p, ob.default_settings(**g["settings"]))
    r = o.solve()
    i = r["info"]
    assert (i["status_val"], i["iterations"], i["oterations"]) == (g["status_val"], g["iterations"], g["oterations"])
    assert close_vec(r["x"], g["x"]) and close_vec(r["y"], g["y"])
    assert [t["kind"] for t in o.trace()] == g["kinds"]
    o.close()
