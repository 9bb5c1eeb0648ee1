"""The N = 1 bench line carries what VERDICT r4 item 6 asks for (reduced mesh: the fields, not the numbers)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_gpu_line_has_first_call_and_hygiene_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nx", "40", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["scaling"] == "strong" and rec["dtype"] == "f64" and rec["vs_baseline"] is None
    cfg, roof = rec["config"], rec["roofline"]
    ph = cfg["first_call_phases_ms"]
    for k in ("dofs_build_ms", "set_forcing_ms", "set_dirichlet_ms", "solver_prepare_ms", "first_init_ms", "first_solve_ms", "mesh_upload_ms_not_counted",
              "ctx_create_ms_not_counted"):
        assert ph[k] > 0, k
    counted = sum(v for k, v in ph.items() if not k.endswith("not_counted"))
    assert abs(cfg["first_call_ms"] - counted) < 1e-6 * counted
    assert cfg["t_setup_ms_untimed"] > 0 and cfg["relres"] <= 1e-10 and cfg["persistent_launch"] == 1
    # algorithmic bytes on the interior block the solve runs on, the full operator's next to them
    n_int, nnz_int = roof["interior_rows"], roof["interior_nnz"]
    assert roof["algorithmic_bytes_per_iteration"] == 12.0 * nnz_int + 4.0 * (n_int + 1) + 16.0 * n_int
    assert roof["algorithmic_bytes_full_operator_per_application"] > roof["algorithmic_bytes_per_iteration"]
    assert "Infinity Cache" in roof["residency"]


def test_secondary_workloads_return_their_fields():
    """extra.c1 (the reference's own fixtures) and the wide run's summary fields at a reduced size"""
    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import capi, meshgen, workloads

    c1 = workloads.run_c1(capi, os.path.join(ROOT, "tests", "golden", "mesh"), reps=5)
    for name, nd in (("unit_square_16", 289), ("unit_square_32", 1089)):
        r = c1[name]
        assert r["dofs"] == nd and r["persistent"] == 1 and r["init_ms"] > 0 and r["solve_ms"] > 0
        assert r["handle_solve_one_column_ms"] > r["handle_solve_per_column_of_64_ms"] > 0
        assert r["max_abs_error_vs_analytic"] < 5e-3
    w = workloads.run_wide(capi, meshgen, nx=40, steps=1, warmup=1)
    assert w["persistent"] == 1 and w["relres"] <= 1e-10 and w["dofs"] == 41**3 and "residency" in w


def test_single_gpu_line_carries_oracle_parity():
    """the CPU leg compares what the oracle computes with what the device path produced for the same workload -- pattern, DOF table,
    boundary DOFs bit-exact, stiff_ / mass_ / force_ entries and the solution within the bars of SURVEY 8(d) -- and the line says so (`parity`).
    Reduced mesh here (the fields and the mechanism; the driver's run carries the full-size figures)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nx", "24", "--cpu-nx", "24", "--steps", "1", "--warmup", "1", "--no-extra"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    par = rec["parity"]
    assert par["ok"] is True
    c3 = par["c3"]
    assert c3["ok"] is True and c3["pattern_equal"] and c3["dof_table_equal"] and c3["boundary_dofs_equal"]
    assert c3["stiff_max_abs_diff_over_max1_Amax"] <= 1e-12 and c3["mass_max_abs_diff_over_max1_Mmax"] <= 1e-12
    assert c3["force_max_abs_diff_over_max1_fmax"] <= 1e-12 and c3["solution_rel_l2"] <= 1e-8
    assert c3["bars"]["matrix_entries"] == 1e-12 and c3["bars"]["solution_rel_l2"] == 1e-8
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["value"] > 0

