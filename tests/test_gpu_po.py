"""GPU parity tests of the pose-graph optimisation (slslam_po_solve) against the oracle and the
committed scipy optimum.  Needs a real MI355X."""
import os

import numpy as np
import pytest

from slslam_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _trace_parity(t0, t1, n=None, tol=1e-8):
    assert len(t0) == len(t1)
    for a, b in list(zip(t0, t1))[:n]:
        assert a["iteration"] == b["iteration"] and a["step_is_successful"] == b["step_is_successful"]
        assert abs(a["cost"] - b["cost"]) <= tol * abs(a["cost"]) + 1e-18
        assert abs(a["trust_region_radius"] - b["trust_region_radius"]) <= 1e-5 * a["trust_region_radius"]


@pytest.mark.parametrize("seed,n,loops", [(1, 12, 1), (2, 40, 3), (3, 75, 4)])
def test_po_matches_oracle(hip, oracle, seed, n, loops):
    """Dual-number Jacobians on the device, dense MFMA Cholesky: cost trace and poses vs the oracle
    (Jet<12> + dense Cholesky).  n = 12/40/75 poses -> 66/234/444 unknowns: one, four and seven
    64-blocks, exercising the ragged last block of the blocked factorisation."""
    g = synth.make_pose_graph(seed, num_poses=n, num_loops=loops)
    x0, s0, t0 = oracle.po_solve(g)
    x1, s1, t1 = hip.po_solve(g)
    _trace_parity(t0, t1, n=3)
    assert s0["num_successful_steps"] == s1["num_successful_steps"]
    assert s0["termination_type"] == s1["termination_type"]
    assert abs(s0["initial_cost"] - s1["initial_cost"]) <= 1e-12 * s0["initial_cost"]
    assert abs(s0["final_cost"] - s1["final_cost"]) <= 1e-7 * s0["final_cost"]
    assert np.abs(x0 - x1).max() < 1e-6
    assert np.array_equal(x1[:6], g["parameters"][:6])              # pose1 of edge 0 is constant (po_problem.cpp:62-63)
    assert s1["num_free_parameters"] == 6 * (n - 1) and s1["num_residual_blocks"] == len(g["pose_index_1"])


def test_po_golden_optimum(hip):
    z = np.load(os.path.join(GOLD, "po_optimum.npz"))
    g = {k: z[k] for k in z.files}
    g["num_poses"] = int(z["num_poses"])
    x, s, t = hip.po_solve(g, max_num_iterations=60, function_tolerance=1e-16, parameter_tolerance=1e-14, gradient_tolerance=1e-16)
    assert abs(s["initial_cost"] - float(z["initial_cost"])) < 1e-12 * float(z["initial_cost"])
    assert abs(s["final_cost"] - float(z["optimum_cost"])) < 1e-8 * float(z["optimum_cost"])
    assert np.abs(x - z["optimum"]).max() < 1e-5


def test_po_consistent_graph_and_edge_cases(hip):
    g = synth.make_pose_graph(4, num_poses=30, num_loops=2)
    truth = g["true_parameters"].reshape(-1, 6)
    cons = []
    for a, b in zip(g["pose_index_1"], g["pose_index_2"]):
        Ra, ta = synth.wt_to_rt(truth[a]); Rb, tb = synth.wt_to_rt(truth[b])
        Rrel = Rb @ Ra.T
        cons.append(synth.rt_to_wt(Rrel, tb - Rrel @ ta))
    g0 = dict(g, constraints=np.array(cons), parameters=g["true_parameters"])
    x, s, t = hip.po_solve(g0)                       # Te = identity on every edge: zero cost, nothing to do
    assert s["initial_cost"] < 1e-25 and np.abs(x - g["true_parameters"]).max() < 1e-12
    # zero iterations / no edges
    x, s, t = hip.po_solve(g, max_num_iterations=0)
    assert np.array_equal(x, g["parameters"])
    ge = dict(num_poses=3, pose_index_1=np.zeros(0, np.int32), pose_index_2=np.zeros(0, np.int32),
              constraints=np.zeros((0, 6)), parameters=np.arange(18.0))
    x, s, t = hip.po_solve(ge)
    assert np.array_equal(x, np.arange(18.0))
    # an unreferenced pose stays untouched; a bad index is rejected
    g2 = dict(g, num_poses=31, parameters=np.concatenate([g["parameters"], np.arange(6.0)]))
    x, s, t = hip.po_solve(g2)
    assert np.array_equal(x[-6:], np.arange(6.0)) and s["num_free_parameters"] == 6 * 29
    bad = dict(g, pose_index_2=g["pose_index_2"].copy()); bad["pose_index_2"][1] = 99
    with pytest.raises(hip.SlslamError):
        hip.po_solve(bad)


def test_po_full_size_loop_closure(hip, oracle):
    """BASELINE config 5: ~260 poses with loop closures, 10 iterations (slam.cpp:1283)."""
    g = synth.make_pose_graph(7, num_poses=260, num_loops=8)
    x1, s1, t1 = hip.po_solve(g)
    assert s1["num_free_parameters"] == 1554
    costs = [r["cost"] for r in t1]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(costs, costs[1:]))
    assert s1["final_cost"] < 0.05 * s1["initial_cost"]
    # loop closure pulls the dead-reckoned trajectory back towards the truth
    c_true = synth.camera_centers(g["true_parameters"])
    e0 = np.linalg.norm(synth.camera_centers(g["parameters"]) - c_true, axis=1)
    e1 = np.linalg.norm(synth.camera_centers(x1) - c_true, axis=1)
    assert np.sqrt((e1 ** 2).mean()) < 0.7 * np.sqrt((e0 ** 2).mean())
    x0, s0, t0 = oracle.po_solve(g)
    _trace_parity(t0, t1, n=3)
    assert abs(s0["final_cost"] - s1["final_cost"]) <= 1e-6 * s0["final_cost"]
    assert np.abs(x0 - x1).max() < 1e-5


def test_po_fp32_factorisation_tolerance(hip):
    """BASELINE config 5 'fp32 vs fp64 tolerance check': the same 260-pose loop-closure graph solved
    with the normal matrix factored in fp32 (v_mfma_f32_16x16x4_f32) and in fp64.  Only the factor is
    single precision (residuals, gradient, cost and LM bookkeeping stay fp64), so the iterates differ
    by the fp32 step error and the solves agree within the STATED tolerance:
    final cost rel 1e-4, poses 1e-4 (rad / m)."""
    g = synth.make_pose_graph(7, num_poses=260, num_loops=8)
    x64, s64, t64 = hip.po_solve(g)
    x32, s32, t32 = hip.po_solve(g, po_factor_fp32=1)
    assert s32["termination_type"] in (0, 2, 3) and s32["num_successful_steps"] >= 1
    assert abs(s32["initial_cost"] - s64["initial_cost"]) <= 1e-13 * s64["initial_cost"]
    assert abs(s32["final_cost"] - s64["final_cost"]) <= 1e-4 * s64["final_cost"]
    d = np.abs(x32 - x64).reshape(-1, 6)
    print("fp32-vs-fp64 factorisation: max |dw| %.3e rad, max |dt| %.3e m, final cost %.9e vs %.9e, steps %d vs %d" % (
        d[:, :3].max(), d[:, 3:].max(), s32["final_cost"], s64["final_cost"],
        s32["num_successful_steps"], s64["num_successful_steps"]))
    assert d.max() < 1e-4
    # small graph: one ragged block
    g2 = synth.make_pose_graph(1, num_poses=12, num_loops=1)
    a, sa, _ = hip.po_solve(g2)
    b, sb, _ = hip.po_solve(g2, po_factor_fp32=1)
    assert np.abs(a - b).max() < 1e-5 and abs(sa["final_cost"] - sb["final_cost"]) <= 1e-5 * sa["final_cost"]


def _add_edges(g, pairs, rng):
    """extra edges between existing poses, constraints from the true poses + a little noise"""
    truth = g["true_parameters"].reshape(-1, 6)
    ed = {(int(a), int(b)): c for a, b, c in zip(g["pose_index_1"], g["pose_index_2"], g["constraints"])}
    for a, b in pairs:
        a, b = (a, b) if a < b else (b, a)
        if (a, b) in ed:
            continue
        Ra, ta = synth.wt_to_rt(truth[a]); Rb, tb = synth.wt_to_rt(truth[b])
        Rrel = synth.rodrigues(rng.normal(0, 1e-3, 3)) @ Rb @ Ra.T
        ed[(a, b)] = synth.rt_to_wt(Rrel, tb - (Rb @ Ra.T) @ ta + rng.normal(0, 2e-3, 3))
    keys = sorted(ed)
    return dict(g, pose_index_1=np.array([k[0] for k in keys], dtype=np.int32), pose_index_2=np.array([k[1] for k in keys], dtype=np.int32),
                constraints=np.array([ed[k] for k in keys]))


@pytest.mark.parametrize("shape", ["single_chain", "ring", "two_ends_one_junction", "hub", "long_chains", "dense_loops", "two_poses", "three_levels", "two_levels_two_paths"])
def test_po_structured_factorisation_topologies(hip, oracle, shape):
    """The default factorisation eliminates chains of poses concurrently and factors only the junction poses
    densely.  Every topology class of the symbolic analysis (free-ended chain, junction-free cycle, chain
    returning to its junction, high-degree junction, chains longer than the cut length, many loop closures)
    must give what the dense factorisation of the whole matrix and the oracle give.  Round 5: long paths are cut on several levels (the cut
    poses of a path are a chain of the next level): the last two shapes."""
    rng = np.random.default_rng(11)
    if shape == "single_chain":
        g = synth.make_pose_graph(21, num_poses=50, num_loops=0)
        g["parameters"] = g["parameters"] + rng.normal(0, 2e-3, g["parameters"].shape) * (np.arange(len(g["parameters"])) >= 6)
    elif shape == "ring":                      # 1..N-1 form a cycle of degree-2 poses (pose 0 is the constant one)
        g = _add_edges(synth.make_pose_graph(22, num_poses=30, num_loops=0), [(1, 29)], rng)
    elif shape == "two_ends_one_junction":     # 5-6-...-14 leaves junction 5 and comes back to it
        g = _add_edges(synth.make_pose_graph(23, num_poses=30, num_loops=0), [(5, 14), (5, 20)], rng)
    elif shape == "hub":
        g = _add_edges(synth.make_pose_graph(24, num_poses=40, num_loops=0), [(10, k) for k in (15, 20, 25, 30, 35, 39)], rng)
    elif shape == "long_chains":
        g = synth.make_pose_graph(25, num_poses=120, num_loops=2)
    elif shape == "dense_loops":
        g = _add_edges(synth.make_pose_graph(26, num_poses=60, num_loops=0), [(i, i + 7) for i in range(1, 50, 3)], rng)
    elif shape == "three_levels":              # (round 5) one path of ~290 poses between two junctions: pieces, chains of cut poses, a chain of THEIR cut poses
        g = _add_edges(synth.make_pose_graph(28, num_poses=300, num_loops=0), [(4, 296), (2, 298)], rng)
        st = hip.po_structure(g)
        assert len(st["chains"]) - st["level1_chains"] >= 7 and max(c[1] for c in st["chains"]) <= 8
    elif shape == "two_levels_two_paths":      # two paths of ~50 poses (two levels each) and short ones around them
        g = _add_edges(synth.make_pose_graph(29, num_poses=120, num_loops=0), [(5, 60), (8, 115), (60, 115)], rng)
        st = hip.po_structure(g)
        assert len(st["chains"]) > st["level1_chains"]
    else:
        g = synth.make_pose_graph(27, num_poses=2, num_loops=0)
        g["parameters"] = g["parameters"] + np.r_[np.zeros(6), rng.normal(0, 1e-2, 6)]
    x0, s0, t0 = oracle.po_solve(g)
    xs, ss, ts = hip.po_solve(g)
    xd, sd, td = hip.po_solve(g, po_dense_factor=1)
    assert np.abs(xs - xd).max() < 1e-9                                  # same system, different elimination order
    assert np.abs(xs - x0).max() < 1e-6
    if shape in ("single_chain", "two_poses"):
        # a tree: every constraint can be met exactly, the solve runs into rounding noise (cost ~ 1e-20) and the
        # iteration at which a tolerance fires is not comparable; the poses are
        assert ss["final_cost"] < 1e-12 * ss["initial_cost"] and s0["final_cost"] < 1e-12 * s0["initial_cost"]
        return
    assert ss["num_successful_steps"] == sd["num_successful_steps"] == s0["num_successful_steps"]
    assert ss["termination_type"] == sd["termination_type"] == s0["termination_type"]
    assert abs(ss["final_cost"] - s0["final_cost"]) <= 1e-7 * s0["final_cost"] + 1e-20
    _trace_parity(t0, ts, n=3)
