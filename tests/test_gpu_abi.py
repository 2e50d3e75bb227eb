"""The drop-in boundary driven from compiled C (no Python, no ctypes mirror): tests/abi_driver.c includes
include/qpdo.h, links against libqpdo_amd.so and replays the call sequence of the reference's MATLAB gateway
(interfaces/mex/qpdo_mex.c:98-281) on the reference's three known-answer QPs (examples/infeasibility_tests.m:30,48,75),
followed by the warm_start / update_bounds / update_q / update_settings re-solve sequence of an MPC caller."""
import subprocess

import pytest

from qpdo_amd import _build

pytestmark = pytest.mark.gpu


def test_c_caller_replays_gateway_sequence_on_known_answers(gpu_required, tmp_path):
    exe = _build.build_abi_driver(str(tmp_path))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "abi driver ok (0 failures)" in out.stdout
    lines = {l.split()[0]: l for l in out.stdout.splitlines() if l and l.split()[0] in ("degenerate", "primal_infeasible", "dual_infeasible")}
    assert "status   1 (solved)" in lines["degenerate"]
    assert "status  -3 (primal infeasible)" in lines["primal_infeasible"]
    assert "status  -4 (dual infeasible)" in lines["dual_infeasible"]
