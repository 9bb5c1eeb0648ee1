"""Two (or N) processes solving C2-size systems on the SAME GPU at the same time: the persistent CG needs all its workgroups resident,
which another process's launch can prevent for a while.  Every solve must still return the right answer -- through the bounded waits
(a launch that cannot get its peers gives up after 50 ms and the solve is re-run through the multi-launch kernels) -- and nothing may hang.
usage: python tools/persist_contention.py [n_procs] [solves]   (rank given internally)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    sys.path.insert(0, ROOT)
    import numpy as np
    from fdapde_loader import load_package
    load_package()
    from fdapde_core_amd import capi, meshgen
    nx, solves = int(sys.argv[2]), int(sys.argv[3])
    nodes, cells, bnd = meshgen.unit_square(nx)
    u_exact, f = meshgen.manufactured(2)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd); nd = c.dofs_build(1)
    c.set_operator(-capi.laplacian()); c.set_forcing(f(c.quadrature_nodes())); c.set_dirichlet(np.zeros(nd)); c.init()
    n_pers, t0 = 0, time.time()
    for _ in range(solves):
        i = c.solve(rtol=1e-10)
        assert i.converged == 1 and i.relres <= 1e-10
        n_pers += i.persistent
        err = np.abs(c.solution() - u_exact(nodes)).max()
        assert err < 1e-4, err
    print(f"worker pid {os.getpid()}: {solves} solves ok, {n_pers} as one persistent launch, {time.time() - t0:.2f} s", flush=True)
    sys.exit(0)
n_procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
solves = int(sys.argv[2]) if len(sys.argv) > 2 else 30
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", "708", str(solves)]) for _ in range(n_procs)]
rc = [p.wait(timeout=600) for p in procs]
print("exit codes", rc)
sys.exit(max(rc))
