"""world_size-2 (and 3) gloo jobs on CPU covering the N > 1 path's host logic: partition, sub-meshes, interface maps,
ownership (P1 and P2: DOFs matched across ranks by node / edge keys), and the distributed Jacobi-PCG and BiCGStab recurrences (tests/dist_worker.py)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,case", [(2, "unit_sphere"), (2, "square"), (3, "cube"), (2, "square:2"), (3, "cube:2"),
                                        (2, "cube:1:adr"), (2, "unit_sphere:2:adr")])
def test_partitioned_pcg_matches_single_domain(world, case):
    port = str(_free_port())
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), port, case],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out}"
        assert f"rank {r}: ok" in out


def test_rank_problems_agree_pairwise():
    """what bench.py's rank 0 ships to the other ranks (dist.rank_problems_p1): every pair of peers lists the DOFs it shares in the SAME
    order (ascending global node id), ownership covers every node exactly once, sub-meshes cover every cell exactly once"""
    import numpy as np

    from fdapde_loader import load_package

    load_package()
    from fdapde_core_amd import dist as fdist
    from fdapde_core_amd import meshgen

    nodes, cells, bnd = meshgen.unit_cube(7)
    for world in (2, 3, 5):
        probs = fdist.rank_problems_p1(nodes, cells, bnd, world)
        assert sum(p["cells"].shape[0] for p in probs) == cells.shape[0]
        owned = np.zeros(nodes.shape[0], dtype=int)
        for p in probs:
            np.add.at(owned, p["l2g"][p["owned"] != 0], 1)
            assert np.array_equal(nodes[p["l2g"]], p["nodes"]) and np.array_equal(p["l2g"][p["cells"]].shape, p["cells"].shape)
        assert np.all(owned == 1)
        for r, p in enumerate(probs):
            for q, peer in enumerate(p["peer_rank"]):
                mine = p["l2g"][p["peer_dof"][p["peer_off"][q]:p["peer_off"][q + 1]]]
                other = probs[peer]
                k = list(other["peer_rank"]).index(r)
                theirs = other["l2g"][other["peer_dof"][other["peer_off"][k]:other["peer_off"][k + 1]]]
                assert np.array_equal(mine, theirs) and np.all(np.diff(mine) > 0)


def test_file_rendezvous_and_self_launch_failure(tmp_path):
    """bench.py's rank bootstrap without a launcher: blobs through a directory; on a box without a HIP device every rank refuses to run
    and the self-launching parent must report that with a non-zero exit code instead of waiting"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    os.environ["FDAPDE_BENCH_RDZV"] = str(tmp_path)
    try:
        a, b = bench.FileRendezvous(0, 2), bench.FileRendezvous(1, 2)
        a.put("id", b"\x00\x01" * 64)
        assert b.get("id") == b"\x00\x01" * 64
        with pytest.raises(TimeoutError):
            b.get("missing", timeout=0.05)
    finally:
        del os.environ["FDAPDE_BENCH_RDZV"]
    from fdapde_loader import load_package

    if load_package().capi.load().fdapde_device_count() >= 1:
        return   # (on a GPU box the launch itself is covered by tests/test_gpu_dist.py)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FDAPDE_BENCH_RDZV")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--nx", "4"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, env=env, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "HIP device" in out.stderr


def test_row_distributed_partition_covers_every_row_completely():
    """fdapde_rowdist_setup's contract, on the host side (dist.node_owners / rowdist_sub_mesh / rowdist_keys_owners): every DOF has exactly
    one owner; the owner's sub-mesh holds EVERY cell touching the DOF (vertex and edge DOFs), so its assembly completes the DOF's row"""
    import numpy as np

    from fdapde_loader import load_package
    from oracle import oracle as o

    load_package()
    from fdapde_core_amd import dist as fdist
    from fdapde_core_amd import meshgen

    for dim, nx in ((2, 9), (3, 5)):
        nodes, cells, bnd = meshgen.unit_square(nx) if dim == 2 else meshgen.unit_cube(nx)
        n = nodes.shape[0]
        all_keys = fdist._cell_keys(cells, n, 2)                     # n_cells x n_basis: the P2 DOFs of every cell by global key
        for world in (2, 3):
            part = fdist.partition_cells(nodes, cells, world)
            owner = fdist.node_owners(cells, part, n, nodes)
            seen = {}
            for r in range(world):
                sub = fdist.rowdist_sub_mesh(nodes, cells, bnd, owner, r)
                m = o.Mesh(sub["nodes"], sub["cells"], sub["boundary"])
                table, _, nd, _ = o.enumerate_dofs(m, 2)
                keys, own = fdist.rowdist_keys_owners(sub, table, owner, n, 2)
                assert keys.size == nd and np.unique(keys).size == nd
                in_sub = np.zeros(cells.shape[0], dtype=bool)
                in_sub[sub["cell_ids"]] = True
                for k in keys[own == r]:
                    assert k not in seen
                    seen[k] = r
                    touching = (all_keys == k).any(axis=1)
                    assert in_sub[touching].all(), "a cell touching an owned DOF is missing from its owner's sub-mesh"
            assert len(seen) == np.unique(all_keys).size
