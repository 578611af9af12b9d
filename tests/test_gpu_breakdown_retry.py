"""GPU: a PCG breakdown under the two-level method or the multigrid never turns a solvable LM step into an invalid one.

Ceres factorises exactly: a step of the reference (`ceres::Solve`, src/PoseGraphSLAM.cpp:1903) is invalid only when the factorisation fails or the model does not
decrease.  libpgo's preconditioners stream fp32 copies of dense inverses, which need not be positive definite at large trust-region
radii; the PCG kernels report r.z < 0, NaN, or — two-level method — a convergence the block-Jacobi part of r.z does not confirm as a BREAKDOWN, and lm_step finishes
the same system with plain block-Jacobi.  PGO_DEBUG_BREAK_COARSE=1 flips the sign of the dense coarse inverse, which forces that path on every system."""
import os

import numpy as np
import pytest

from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util

pytestmark = pytest.mark.gpu


def solve(g, switchable, **kw):
    q, t, s = util.initial_state(g, switchable)
    P = util.pgo_problem(g, switchable, **kw)
    out = P.solve(q, t, s)
    P.close()
    return out


@pytest.fixture
def broken_coarse_inverse():
    os.environ["PGO_DEBUG_BREAK_COARSE"] = "1"
    yield
    os.environ.pop("PGO_DEBUG_BREAK_COARSE", None)


def trajectory(sm):
    # (a step the plain run throws away at an early-rejection pause is rejected at full accuracy by the retried run: the same decision)
    same = {capi.STEP_REJECTED_AT_PAUSE: capi.STEP_REJECTED_RHO}
    return [(sm.iterations[k].step_is_valid, sm.iterations[k].step_is_successful, same.get(sm.iterations[k].reason, sm.iterations[k].reason)) for k in range(sm.num_logged)]


@pytest.mark.parametrize("name,switchable", [("C1F5", True), ("S2500", False), ("G9000", True)])
def test_forced_breakdown_is_retried_with_block_jacobi_and_the_trajectory_is_the_plain_one(name, switchable, broken_coarse_inverse):
    if name == "S2500":
        g = graphgen.generate(2500, 250, odom_f_max=1, seed=2, outlier_frac=0.0)             # two-level method, plain loops (C2's structure)
    elif name == "G9000":
        g = graphgen.generate(9000, 9000, odom_f_max=2, seed=3)                               # multigrid (>= mg_min_keyframes_switchable)
    else:
        g = graphgen.config(name)
    kw = dict(mg_switch_iterations=0) if name == "G9000" else {}                              # (the multigrid on every system, so that every system breaks down)
    _, tb, sb, broken = solve(g, switchable, **kw)
    os.environ.pop("PGO_DEBUG_BREAK_COARSE", None)
    _, tp, sp, plain = solve(g, switchable, coarse_aggregates=0, mg_min_keyframes=0)
    assert broken.pcg_retries > 0
    retried = [k for k in range(1, broken.num_logged) if broken.iterations[k].preconditioner & capi.PRECOND_RETRIED]
    assert len(retried) == broken.pcg_retries
    assert trajectory(broken) == trajectory(plain), (trajectory(broken), trajectory(plain))
    assert all(broken.iterations[k].step_is_valid for k in range(broken.num_logged))
    for k in range(plain.num_logged):
        assert abs(broken.iterations[k].cost - plain.iterations[k].cost) <= 1e-9 * plain.iterations[k].cost, k
    assert np.abs(tb - tp).max() <= 1e-6


def test_reason_codes_name_every_way_a_step_can_end():
    """accepted / rejected by rho / rejected at a pause / converged on C3-structured graphs; the invalid ones are in tests/test_gpu_failure_contract.py"""
    g = graphgen.generate(20000, 20000, odom_f_max=2, seed=3)
    _, _, _, sm = solve(g, True, max_num_iterations=40)
    seen = {sm.iterations[k].reason for k in range(1, sm.num_logged)}
    for k in range(1, sm.num_logged):
        it = sm.iterations[k]
        if it.reason == capi.STEP_ACCEPTED:
            assert it.step_is_valid and it.step_is_successful and it.relative_decrease > 1e-3
        elif it.reason in (capi.STEP_REJECTED_RHO, capi.STEP_REJECTED_AT_PAUSE):
            assert it.step_is_valid and not it.step_is_successful
        elif it.reason == capi.STEP_CONVERGED:
            assert k == sm.num_logged - 1 and sm.termination_type == capi.CONVERGENCE
    assert capi.STEP_ACCEPTED in seen and capi.STEP_REJECTED_AT_PAUSE in seen and capi.STEP_CONVERGED in seen, seen
    _, _, _, sm2 = solve(g, True, max_num_iterations=12, cg_early_tolerance=0.0, cg_mid_tolerance=0.0)
    assert capi.STEP_REJECTED_RHO in {sm2.iterations[k].reason for k in range(1, sm2.num_logged)}
