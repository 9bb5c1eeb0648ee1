"""C5's operator is almost symmetric (cell Peclet 0.007, domain Peclet ~1): BiCGStab pays two operator applications per iteration where a method with a
truncated long recurrence -- ORTHOMIN(k) / GCR(k): residual minimised over the last k directions -- pays one.  Both in numpy on C5's Jacobi-scaled interior
block at reduced sizes, operator applications to rtol 1e-10, several right-hand sides differing in the last bits.  usage: c5_orthomin_proto.py [nx ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from c5_idr_proto import build, bicgstab   # noqa: E402


def orthomin(A, b, k, tol, maxit=40000):
    x, r, nb = np.zeros_like(b), b.copy(), np.linalg.norm(b)
    P, AP, nrm = [], [], []
    mv = 0
    while mv < maxit:
        Ar = A @ r
        mv += 1
        p, Ap = r.copy(), Ar.copy()
        for pi, Api, ni in zip(P, AP, nrm):
            beta = -(Ar @ Api) / ni
            p += beta * pi
            Ap += beta * Api
        nn = Ap @ Ap
        alpha = (r @ Ap) / nn
        x += alpha * p
        r -= alpha * Ap
        if np.linalg.norm(r) <= tol * nb:
            return x, mv
        P.append(p), AP.append(Ap), nrm.append(nn)
        if len(P) > k:
            P.pop(0), AP.pop(0), nrm.pop(0)
    return x, -mv


if __name__ == "__main__":
    for nx in [int(a) for a in sys.argv[1:]] or (12, 20, 28):
        t0 = time.time()
        A, b = build(nx)
        rng = np.random.default_rng(3)
        rows = []
        for j in range(3):
            bj = b if j == 0 else b * (1.0 + 2e-16 * rng.integers(-2, 3, b.size))
            _, m1 = bicgstab(A, bj, 1e-10)
            res = [m1]
            for k in (1, 2, 4, 8):
                xk, mk = orthomin(A, bj, k, 1e-10)
                res.append(mk)
                tr = np.linalg.norm(bj - A @ xk) / np.linalg.norm(bj)
            rows.append((res, tr))
        print(f"nx {nx}, {b.size} interior DOFs: applications BiCGStab | ORTHOMIN(1) (2) (4) (8), three right-hand sides: "
              + "; ".join(f"{r[0]} | {r[1]} {r[2]} {r[3]} {r[4]}" for r, _ in rows) + f"  (true relres of the last {rows[-1][1]:.1e}) [{time.time() - t0:.0f} s]", flush=True)
