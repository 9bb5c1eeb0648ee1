#!/usr/bin/env python3
"""Single-launch solves: one knob of fdapde_tune at several values on the same context, interleaved (value order repeated `--rounds` times),
3-D P1 Laplacian (CG) over a list of sizes; nx 119 is C3.  Prints iterations and microseconds per iteration inside the launch.

  python tools/persist_knob_ab.py --knob persist_prefetch --values 0,1,2 --nx 72,100,119"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fdapde_loader import load_package

capi = load_package().capi
from fdapde_core_amd import meshgen   # noqa: E402


def run(dim, nx, knob, values, rounds, order, adr, fixed):
    nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
    _, f = meshgen.manufactured(dim)
    c = capi.Context(0)
    c.mesh_upload(nodes, cells, bnd)
    nd = c.dofs_build(order)
    op = -capi.laplacian()
    if adr:
        op = op + capi.advection(np.array([1.0, 0.5, 0.25])[:dim]) + capi.reaction(1.0)
    c.set_operator(op)
    c.set_forcing(f(c.quadrature_nodes()))
    c.set_dirichlet(np.zeros(nd))
    c.init()
    for k, v in fixed:
        c.tune(k, v)
    res = {v: [] for v in values}
    lay = None
    for _ in range(rounds):
        for v in values:
            c.tune(knob, v)
            c.solve(rtol=1e-10)
            i = c.solve(rtol=1e-10)
            lay = c.solver_layout_kind(True)
            res[v].append((i.iters, 1e3 * i.launch_ms / max(i.iters, 1), i.persistent, lay))
    def lay_txt(la):
        return f"G={la['workgroups']} R={la['rows_per_thread']} kind={la['kind']} sym={la['sym']}"
    txt = "  ".join(f"{knob}={v} [{lay_txt(res[v][0][3])}]: {res[v][0][0]} it " + "/".join(f"{us:.2f}" for _, us, _, _ in res[v]) + " us/it" +
                    ("" if res[v][0][2] else " (multi-launch)") for v in values)
    print(f"{dim}-D P{order}{' ADR' if adr else ''} nx {nx}: {nd} DOFs  {txt}", flush=True)
    c.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--knob", required=True)
    ap.add_argument("--values", default="0,1")
    ap.add_argument("--nx", default="72,100,119")
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--adr", action="store_true", help="advection-diffusion-reaction operator (BiCGStab)")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--set", default="", help="other knobs set once, key=value,key=value")
    a = ap.parse_args()
    for nx in (int(t) for t in a.nx.split(",")):
        run(a.dim, nx, a.knob, [int(t) for t in a.values.split(",")], a.rounds, a.order, a.adr,
            [(kv.split("=")[0], int(kv.split("=")[1])) for kv in a.set.split(",") if kv])
