"""BiCGStab(2) (Sleijpen & Fokkema 1993) next to BiCGStab on C5's Jacobi-scaled interior block at reduced sizes (numpy on the host, the matrix assembled by the
device): operator applications to rtol 1e-10 for the published right-hand side and for copies of it perturbed in the last bits -- does the longer recurrence
need fewer applications, and does its count move less?  usage: c5_bicgstab2_proto.py [nx ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from c5_idr_proto import build, bicgstab   # noqa: E402


def bicgstab_l(A, b, ell, tol, maxit=40000):
    n = b.size
    x = np.zeros(n)
    r = [b.copy()] + [np.zeros(n) for _ in range(ell)]
    u = [np.zeros(n) for _ in range(ell + 1)]
    rt = b.copy()
    rho0, alpha, omega, mv, nb = 1.0, 0.0, 1.0, 0, np.linalg.norm(b)
    while mv < maxit:
        rho0 = -omega * rho0
        for j in range(ell):   # the BiCG part
            rho1 = r[j] @ rt
            beta = alpha * rho1 / rho0
            rho0 = rho1
            for i in range(j + 1):
                u[i] = r[i] - beta * u[i]
            u[j + 1] = A @ u[j]
            gamma = u[j + 1] @ rt
            alpha = rho0 / gamma
            for i in range(j + 1):
                r[i] = r[i] - alpha * u[i + 1]
            r[j + 1] = A @ r[j]
            mv += 2
            x = x + alpha * u[0]
        # the minimal-residual part: modified Gram-Schmidt on r_1 .. r_ell
        tau = np.zeros((ell + 1, ell + 1))
        sigma, gp = np.zeros(ell + 1), np.zeros(ell + 1)
        for j in range(1, ell + 1):
            for i in range(1, j):
                tau[i, j] = (r[j] @ r[i]) / sigma[i]
                r[j] = r[j] - tau[i, j] * r[i]
            sigma[j] = r[j] @ r[j]
            gp[j] = (r[0] @ r[j]) / sigma[j]
        g = np.zeros(ell + 1)
        g[ell] = gp[ell]
        omega = g[ell]
        for j in range(ell - 1, 0, -1):
            g[j] = gp[j] - sum(tau[j, i] * g[i] for i in range(j + 1, ell + 1))
        gpp = np.zeros(ell + 1)
        for j in range(1, ell):
            gpp[j] = g[j + 1] + sum(tau[j, i] * g[i + 1] for i in range(j + 1, ell))
        x = x + g[1] * r[0]
        r[0] = r[0] - gp[ell] * r[ell]
        u[0] = u[0] - g[ell] * u[ell]
        for j in range(1, ell):
            u[0] = u[0] - g[j] * u[j]
            x = x + gpp[j] * r[j]
            r[0] = r[0] - gp[j] * r[j]
        if np.linalg.norm(r[0]) <= tol * nb:
            return x, mv
    return x, -mv


if __name__ == "__main__":
    for nx in [int(a) for a in sys.argv[1:]] or (12, 20, 28):
        t0 = time.time()
        A, b = build(nx)
        rng = np.random.default_rng(3)
        rows = []
        for k in range(5):
            bk = b if k == 0 else b * (1.0 + 2e-16 * rng.integers(-2, 3, b.size))
            x1, m1 = bicgstab(A, bk, 1e-10)
            x2, m2 = bicgstab_l(A, bk, 2, 1e-10)
            x4, m4 = bicgstab_l(A, bk, 4, 1e-10)
            tr = [np.linalg.norm(bk - A @ x) / np.linalg.norm(bk) for x in (x1, x2, x4)]
            rows.append((m1, m2, m4, max(tr)))
        print(f"nx {nx}, {b.size} interior DOFs: applications BiCGStab / BiCGStab(2) / BiCGStab(4) over 5 right-hand sides differing in the last bits: "
              + ", ".join(f"{a}/{c}/{d}" for a, c, d, _ in rows) + f"; worst true relres {max(r[3] for r in rows):.1e} [{time.time() - t0:.0f} s]", flush=True)
