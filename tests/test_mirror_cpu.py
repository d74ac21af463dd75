"""CPU-only checks of the host-side mirror: config contract (field names, defaults, validators,
JSON round trip) and that the mirror modules expose the reference's names."""
import json

import pytest


@pytest.fixture(scope="module")
def V():
    import vch_amd
    vch_amd.build()
    return vch_amd


def test_config_contract_2d(V, tmp_path):
    K2 = V.module("Vch_control_2D.config")
    f, o = K2.ForwardSolverConfig(), K2.OptimizationConfig()
    assert (f.Nx, f.Ny, f.Lx, f.Ly, f.T, f.dt_initial, f.tau, f.gamma, f.c1, f.c2) == (128, 128, 1.0, 1.0, 1.0, 1e-2, 0.05, 10.0, 0.75, 1.0)
    assert f.kappa == 0.01 ** 2
    assert (o.b1, o.b2, o.b3, o.kappa_sparsity, o.alpha_max, o.max_iter, o.u_min, o.u_max) == (5.0, 10.0, 1e-4, 1e-4, 50.0, 500, -1.0, 1.0)
    with pytest.raises(Exception):
        K2.ForwardSolverConfig(c1=1.0, c2=0.9)          # c2 must exceed c1 (K2:115-120)
    with pytest.raises(Exception):
        K2.OptimizationConfig(u_min=1.0, u_max=0.0)     # K2:146-150
    with pytest.raises(Exception):
        K2.ForwardSolverConfig(Nx=10)                   # gt=10
    p = tmp_path / "cfg.json"
    K2.save_params(K2.ForwardSolverConfig(Nx=64), o, 7, str(p))
    d = json.load(open(p))
    assert set(d) == {"forward_solver", "optimization", "last_run_iterations"} and d["last_run_iterations"] == 7
    back = K2.load_params(str(p))
    assert back.forward_solver.Nx == 64
    assert K2.load_params(str(tmp_path / "missing.json")).forward_solver.Nx == 128


def test_config_contract_1d(V):
    K1 = V.module("Vch_control_1D.config")
    f, o = K1.ForwardSolverConfig(), K1.OptimizationConfig()
    assert (f.N, f.kappa) == (128, 0.03 ** 2)
    assert (o.b1, o.b2, o.b3, o.kappa_sparsity, o.alpha_max, o.max_iter) == (0.3, 13.0, 0.0019, 0.00009, 100.0, 1000)


def test_mirror_exposes_reference_names(V):
    names = {
        "Vch_control_2D.Forward2_solver": ["regularized_log", "laplacian_matrix_neumann", "apply_laplacian", "initialize_mu",
                                           "solve_w", "solve_mu_residual", "solve_phi_residual", "assemble_jacobian",
                                           "free_energy", "newton_raphson", "trapz_weights", "init_phi_random",
                                           "run_main_simulation", "instability_report"],
        "Vch_control_2D.backward2_solver": ["fpp_log", "run_backward"],
        "Vch_control_2D.cost2_and_function": ["calculate_cost", "calculate_gradient", "proximal_step"],
        "Vch_control_2D.GD2_configured": ["perform_backtracking_line_search_2D", "build_targets"],
        "Vch_control_2D.second_order_conditions_2d": ["approximate_second_order_condition_2d", "verify_sparsity_condition",
                                                      "_generate_direction"],
        "Vch_control_1D.second_order_conditions": ["approximate_second_order_condition", "_generate_direction"],
        "Vch_control_1D.Forward_solver": ["regularized_log", "laplacian_matrix_neumann", "apply_laplacian", "initialize_mu",
                                          "solve_w", "solve_mu_residual", "solve_phi_residual", "assemble_jacobian",
                                          "newton_raphson", "trapz_weights", "free_energy", "init_phi_random",
                                          "run_main_simulation"],
        "Vch_control_1D.backward_solver": ["fpp_log", "run_backward"],
        "Vch_control_1D.cost_and_function": ["calculate_cost", "calculate_gradient", "perform_gradient_step"],
        "Vch_control_1D.GD_1D": ["perform_proximal_and_projection", "perform_backtracking_line_search",
                                 "verify_sparsity_condition", "build_targets_1d"],
    }
    for mod, fns in names.items():
        m = V.module(mod)
        for f in fns:
            assert callable(getattr(m, f)), (mod, f)


def test_host_helpers_match_oracle(V):
    import numpy as np
    from oracle import vch2d_oracle as O2, vch1d_oracle as O1
    F2 = V.module("Vch_control_2D.Forward2_solver")
    F1 = V.module("Vch_control_1D.Forward_solver")
    assert np.array_equal(F2.init_phi_random(20, 14, 1e-2, amp=0.1, seed=5), O2.init_phi_random(20, 14, 1e-2, amp=0.1, seed=5))
    assert np.array_equal(F1.init_phi_random(50, 1e-2, amp=0.01, seed=5), O1.init_phi_random(50, 1e-2, amp=0.01, seed=5))
    phi = np.random.default_rng(0).uniform(-0.99, 0.99, (9, 7))
    assert np.array_equal(F2.regularized_log(phi, 1e-2), O2.reg_log(phi))


def test_direction_generators_match_reference_goldens(V):
    """The critical-cone direction generators are host logic (NumPy RNG stream): bit-exact."""
    import numpy as np
    from conftest import golden
    S2 = V.module("Vch_control_2D.second_order_conditions_2d")
    S1 = V.module("Vch_control_1D.second_order_conditions")
    gp, gs = golden("g2d_pgd_16.npz"), golden("g2d_soc_16.npz")
    assert np.array_equal(S2._generate_direction(gp["u_final"], gs["r_opt"], -1.0, 1.0, np.random.default_rng(42)), gs["h0"])
    assert np.array_equal(S2._generate_direction(gs["u_sat"], gs["r_opt"], -1.0, 1.0, np.random.default_rng(7)), gs["h_sat"])
    gp, gs = golden("g1d_pgd_32.npz"), golden("g1d_soc_32.npz")
    assert np.array_equal(S1._generate_direction(gp["u_final"], gs["r_opt"], -1.0, 1.0, 0.00009, 0.0019,
                                                 np.random.default_rng(42)), gs["h0"])
