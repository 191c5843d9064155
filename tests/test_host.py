"""CPU tests of the host logic: C-ABI library loads and exports every declared symbol (no compute without a
GPU), index helpers, cyclic layout round trips, grid rule, and the N>1 plumbing under gloo (world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "eigenexa_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(eigx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ctypes

    from eigenexa_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/eigenexa_amd.h but not exported"
    # and the ctypes table covers the header one to one
    assert set(names) == set(_lib.SIGNATURES.keys())


def test_product_does_not_touch_the_oracle():
    """the shipped package must not import, link or call anything under oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "eigenexa_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".F90")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.replace("orchestr", ""), f"{f} mentions the oracle"
    out = subprocess.run(["nm", "-D", os.path.join(ROOT, "eigenexa_amd", "lib", "libeigenexa_amd.so")],
                         capture_output=True, text=True).stdout
    assert "orc_" not in out


def test_no_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from eigenexa_amd import _lib

    lib = _lib.load()
    assert lib.eigx_init(0) == -4  # EIGX_ERR_NO_DEVICE: there is no CPU path
    import eigenexa_amd as ee

    with pytest.raises(RuntimeError):
        ee.eigen_init()


def test_index_helpers_match_reference_formulas():
    """src/eigen_libs0.F:1825 (loop_start), :1911 (loop_end), :1995, :2079, :2163, :2247; SURVEY appendix A"""
    from eigenexa_amd import _lib

    lib = _lib.load()
    for P in (1, 2, 3, 4, 7):
        for p in range(1, P + 1):
            owned = [g for g in range(1, 60) if (g - 1) % P + 1 == p]
            for g in range(1, 60):
                ls = lib.eigx_loop_start(g, P, p)   # first local index whose global >= g
                le = lib.eigx_loop_end(g, P, p)     # last local index whose global <= g
                assert ls == 1 + sum(1 for x in owned if x < g)
                assert le == sum(1 for x in owned if x <= g)
                assert lib.eigx_owner_node(g, P, p) == (g - 1) % P + 1
                assert lib.eigx_translate_g2l(g, P, p) == (g - 1) // P + 1
                oi = lib.eigx_owner_index(g, P, p)
                assert oi == ((g - 1) // P + 1 if g in owned else -1)
            for l in range(1, 10):
                assert lib.eigx_translate_l2g(l, P, p) == (l - 1) * P + p


def test_grid_rule():
    """src/eigen_libs0.F:526-570: 1->1x1, 2->1x2, 4->2x2, 8->2x4, 6->2x3, 7->1x7"""
    from eigenexa_amd import layout

    assert [layout.grid_shape(p) for p in (1, 2, 4, 8, 6, 7, 16)] == [(1, 1), (1, 2), (2, 2), (2, 4), (2, 3),
                                                                     (1, 7), (4, 4)]
    assert layout.rank_coords(5, 8) == (1, 2)         # column-major: x = r % Px, y = r // Px
    assert layout.rank_coords(5, 8, "R") == (1, 1)


@pytest.mark.parametrize("nranks", [1, 2, 4, 6, 8])
def test_cyclic_scatter_gather_roundtrip(nranks):
    from eigenexa_amd import layout

    n = 37
    A = layout.random_symmetric(n)
    blocks = [layout.scatter_cyclic(A, nranks, r) for r in range(nranks)]
    assert np.array_equal(layout.gather_cyclic(blocks, n, n), A)
    # generators evaluated at local index sets give the same local blocks (layout independence)
    Px, Py = layout.grid_shape(nranks)
    for r in range(nranks):
        px, py = layout.rank_coords(r, nranks)
        loc = layout.random_symmetric(n, rows=np.arange(px, n, Px), cols=np.arange(py, n, Py))
        assert np.array_equal(loc, blocks[r][: loc.shape[0], : loc.shape[1]])


def test_torch_generator_is_bit_identical():
    from eigenexa_amd import layout

    n = 131
    assert np.array_equal(layout.random_symmetric(n), layout.random_symmetric_torch(n, "cpu", chunk=50).numpy())
    rows, cols = np.arange(1, n, 2), np.arange(0, n, 4)
    assert np.array_equal(layout.random_symmetric(n, rows=rows, cols=cols),
                          layout.random_symmetric_torch(n, "cpu", rows=rows, cols=cols).numpy())


def test_frank_formula():
    from eigenexa_amd import layout

    n = 50
    assert np.abs(np.linalg.eigvalsh(layout.frank(n)) - layout.frank_eigenvalues(n)).max() < 1e-9


_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from eigenexa_amd import layout
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
n = 29
A = layout.random_symmetric(n)
mine = layout.scatter_cyclic(A, 2, rank)
# 1) unique-id style broadcast used by eigen_init(comm): rank 0's 128 bytes reach everyone
t = torch.arange(128, dtype=torch.uint8) if rank == 0 else torch.zeros(128, dtype=torch.uint8)
dist.broadcast(t, src=0)
assert bytes(t.tolist()) == bytes(range(128))
# 2) gather the cyclic blocks and rebuild the global matrix on every rank
Px, Py = layout.grid_shape(2)
blocks = [torch.zeros(n, (n + 1) // 2, dtype=torch.float64) for _ in range(2)]
pad = torch.zeros(n, (n + 1) // 2, dtype=torch.float64)
pad[: mine.shape[0], : mine.shape[1]] = torch.from_numpy(np.ascontiguousarray(mine))
dist.all_gather(blocks, pad)
G = layout.gather_cyclic([b.numpy() for b in blocks], n, n)
assert np.array_equal(G, A)
# 3) the row-group / column-group sums of a distributed mat-vec reproduce A @ u (the reduction's pattern)
u = np.arange(1, n + 1, dtype=np.float64)
px, py = layout.rank_coords(rank, 2)
part = np.zeros(n)
part[px::Px] = mine[: layout.local_count(n, px, Px), : layout.local_count(n, py, Py)] @ u[py::Py]
tt = torch.from_numpy(part)
dist.all_reduce(tt)
assert np.allclose(tt.numpy(), A @ u)
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
'''


def test_gloo_world_size_2(tmp_path):
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT, "port": port})
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"OK {r}" in o, o


def test_benchmark_driver_input_format(tmp_path):
    """the reference driver's input-file format (benchmark/main2.f:262-300, benchmark/IN): comments, 8 integers
    per case, a non-positive N ends the run"""
    from eigenexa_amd import benchmark

    p = tmp_path / "IN"
    p.write_text("! N nvec bx by m t s e\n 1000     0 48 128 1 0 1 0\n!comment\n 64 64 48 128 2 2 0 1\n-1 0 0 0 0 0 0 0\n"
                 " 5 5 48 128 1 0 0 1\n")
    cases = list(benchmark.parse_input(str(p)))
    assert cases == [(1000, 0, 48, 128, 1, 0, 1, 0), (64, 64, 48, 128, 2, 2, 0, 1)]
    assert benchmark.MODES[0] == "N" and benchmark.MODES[2] == "X" and benchmark.MODES[5] == "C"
    assert benchmark._verdict(1e-9) == "PASSED" and benchmark._verdict(1e-5) == "CAUTION"
    assert benchmark._verdict(1e-3) == "FAILED"
    msgs = []
    lam = np.array([1.0, 2.0, 3.0])
    assert benchmark.w_test(lam * (1 + 1e-12), lam, msgs.append)
    assert any("Relative Error *** : PASSED" in m for m in msgs)


def test_numroc_and_block_cyclic_layout():
    """NUMROC of the C-ABI against the python restatement and a brute-force count; block-cyclic scatter/gather round
    trip (the layout eigx_solve_bc accepts); nb = 1 reduces to the cyclic layout of the EigenExa API"""
    from eigenexa_amd import _lib, layout

    lib = _lib.load()
    for n in (1, 5, 64, 100, 301):
        for nb in (1, 2, 7, 32, 64):
            for P in (1, 2, 3, 4):
                tot = 0
                for p in range(P):
                    c = layout.numroc(n, nb, p, P)
                    assert c == lib.eigx_numroc(n, nb, p, P) == len(layout.block_cyclic_indices(n, nb, p, P))
                    if nb == 1:
                        assert c == layout.local_count(n, p, P)
                    tot += c
                assert tot == n
    assert lib.eigx_numroc(10, 0, 0, 2) == -1 and lib.eigx_numroc(10, 2, 2, 2) == -1
    A = np.arange(35 * 29.0).reshape(35, 29)
    for nb in (1, 4, 16):
        blocks = [layout.scatter_block_cyclic(A, nb, 4, r) for r in range(4)]
        assert (layout.gather_block_cyclic(blocks, 35, 29, nb) == A).all()
    # nb = 1 block-cyclic == cyclic
    for r in range(4):
        assert (layout.scatter_block_cyclic(A, 1, 4, r) == layout.scatter_cyclic(A, 4, r)).all()
