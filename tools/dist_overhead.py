"""Per-iteration cost of the multi-GPU code path on ONE rank (1-rank RCCL communicator, artificial interface set of the size an
8-way partition of C3 has): SpMV + pack + ncclAllReduce + single-reduction update against the plain single-GPU iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdapde_loader import load_package
load_package()
from fdapde_core_amd import capi, meshgen
nx = int(os.environ.get("NX", "119")); n_if = int(os.environ.get("NIF", "90000"))
ctx = capi.Context(0)
ctx.mesh_upload(*meshgen.unit_cube(nx)); nd = ctx.dofs_build(1)
u_exact, f = meshgen.manufactured(3)
ctx.set_operator(-capi.laplacian()); ctx.set_forcing(f(ctx.quadrature_nodes())); ctx.set_dirichlet(np.zeros(nd)); ctx.init()
for _ in range(2):
    a = ctx.solve(rtol=1e-10)
print(f"plain (fused CG)      : {a.iters} iterations, {a.t_solve_ms:.2f} ms = {a.t_solve_ms / a.iters * 1e3:.1f} us / iteration")
ctx.tune("persist", 0)   # the multi-launch kernels on the compact solver pattern, as the partitioned path uses them
b = ctx.solve(rtol=1e-10, method=capi.SOLVER_CG_SR)
b = ctx.solve(rtol=1e-10, method=capi.SOLVER_CG_SR)
print(f"single-reduction CG   : {b.iters} iterations, {b.t_solve_ms:.2f} ms = {b.t_solve_ms / b.iters * 1e3:.1f} us / iteration")
ctx.comm_init(1, 0, capi.Context.comm_unique_id())
local = np.sort(np.random.default_rng(1).choice(nd, size=n_if, replace=False)).astype(np.int32)
ctx.halo_setup(n_if, local, np.arange(n_if, dtype=np.int32), np.ones(nd, dtype=np.uint8))
for _ in range(2):   # world > 1 selects the single-reduction form; on one rank it has to be asked for
    d = ctx.solve(rtol=1e-10, method=capi.SOLVER_CG_SR)
print(f"1-rank RCCL, n_if {n_if}: {d.iters} iterations, {d.t_solve_ms:.2f} ms = {d.t_solve_ms / d.iters * 1e3:.1f} us / iteration (method {d.method_used})")
# neighbour-only exchange (fdapde_halo_setup_peers): on ONE rank there is no peer, what remains per iteration is the pack / sum launches
# on empty lists and the grouped RCCL call with the all-reduce of the two scalars
ctx.halo_setup_peers(np.zeros(0, dtype=np.int32), np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.ones(nd, dtype=np.uint8))
for _ in range(2):
    e = ctx.solve(rtol=1e-10, method=capi.SOLVER_CG_SR)
print(f"1-rank RCCL, neighbour-only exchange (no peer): {e.iters} iterations, {e.t_solve_ms:.2f} ms = {e.t_solve_ms / e.iters * 1e3:.1f} us / iteration "
      f"(overhead over the single-reduction CG {(e.t_solve_ms / e.iters - b.t_solve_ms / b.iters) * 1e3:.1f} us; dense form {(d.t_solve_ms / d.iters - b.t_solve_ms / b.iters) * 1e3:.1f} us)")
