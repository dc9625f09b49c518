"""-m gpu: device generators / serialize / triangle masks vs the oracle (bit-exact: integer LCG + one division)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import capital_oracle as orc  # noqa: E402
from tests.gpu_util import DEV, to_dev, to_host  # noqa: E402


@pytest.mark.parametrize("n,d", [(64, 1), (100, 1), (37, 2), (64, 2), (101, 3), (256, 4)])
def test_distribute_symmetric_bit_exact(n, d):
    from capital_amd.matrix import matrix
    for x in range(d):
        for y in range(d):
            A = matrix(n, n, d, d)
            A.distribute_symmetric(x, y, d, d, 0, True)
            assert np.array_equal(A.to_numpy(), orc.symmetric_local(n, x, y, d, True))


def test_distribute_symmetric_matches_reference_dump(golden_dir):
    import os
    from capital_amd.matrix import matrix
    g = np.load(os.path.join(golden_dir, "cholinv_n100_ci0_s2_bc-4.npz"))
    A = matrix(100, 100, 1, 1); A.distribute_symmetric(0, 0, 1, 1, 0, True)
    assert np.array_equal(A.to_numpy(), g["A"])      # the REAL reference's generator output


def test_distribute_symmetric_seed_wrap():
    """N^2 > 2^32: seeds wrap mod 2^32 exactly like srand48 - checked on the last rows of N = 70000."""
    from capital_amd import _lib
    from capital_amd._util import cur_stream
    n, d = 70000, 8          # local piece 8750 x 8750 on grid position (7, 7)
    nl = orc.local_dim(n, d)
    buf = torch.empty(nl, nl, dtype=torch.float64, device=DEV)
    _lib.check(_lib.lib().cap_fill_symmetric(buf.data_ptr(), nl, n, 7, 7, d, 1, cur_stream()))
    got = buf.t()[-3:, -3:].cpu().numpy()
    gy = 7 + (np.arange(nl - 3, nl)) * d; gx = gy.copy()
    ref = np.zeros((3, 3))
    for i, r in enumerate(gy):
        for j, c in enumerate(gx):
            if r < n and c < n:
                ref[i, j] = float(orc.drand48_of_seed(max(r, c) + n * min(r, c))) + (n if r == c else 0)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("m,n,dx,dy,key", [(256, 16, 1, 1, 0), (192, 12, 1, 1, 0), (1000, 8, 1, 4, 3), (37, 5, 1, 2, 1), (64, 6, 2, 2, 7)])
def test_distribute_random_bit_exact(m, n, dx, dy, key):
    from capital_amd.matrix import matrix
    for x in range(dx):
        for y in range(dy):
            A = matrix(n, m, dx, dy)
            A.distribute_random(x, y, dx, dy, key)
            assert np.array_equal(A.to_numpy(), orc.random_local(m, n, x, y, dx, dy, key))


def test_serialize_packed_roundtrip_and_windows():
    from capital_amd.matrix import serialize
    n = 50
    a = np.triu(np.random.default_rng(0).standard_normal((n, n)))
    A, Av = to_dev(a)
    packed = torch.zeros(n * (n + 1) // 2, dtype=torch.float64, device=DEV)
    serialize(A, packed, (0, n, 0, n), (0, 0), tri_only=True, src_ld=n, dst_packed=True)
    assert np.array_equal(packed.cpu().numpy(), orc.pack_upper(a))            # uppertri::_offset, structure.h:39
    B = torch.full((n, n), 3.0, dtype=torch.float64, device=DEV)
    serialize(packed, B, (0, n, 0, n), (0, 0), tri_only=True, zero_lower=True, src_packed=True, dst_ld=n)
    assert np.array_equal(to_host(B.t()), a)
    # rect window copy (serialize<rect,rect> with offsets, as cholinv.hpp:122)
    W = torch.zeros((20, 30), dtype=torch.float64, device=DEV)                # 30 x 20 column-major, ld 30
    serialize(A, W, (5, 25, 10, 30), (3, 0), src_ld=n, dst_ld=30)
    assert np.array_equal(to_host(W.t())[3:23, :20], a[5:25, 10:30])


@pytest.mark.parametrize("d", [1, 2, 3])
def test_remove_triangle(d):
    from capital_amd import _lib
    from capital_amd._util import cur_stream
    n = 23
    full = np.random.default_rng(d).standard_normal((n, n))
    for x in range(d):
        for y in range(d):
            loc = orc.cyclic_local(full, x, y, d, d)
            L, Lv = to_dev(loc)
            _lib.check(_lib.lib().cap_remove_triangle(L.data_ptr(), loc.shape[0], loc.shape[0], loc.shape[1], x, y, d, 1, cur_stream()))
            assert np.array_equal(to_host(Lv), orc.cyclic_local(np.triu(full), x, y, d, d))


@pytest.mark.parametrize("m,n,dx,dy", [(64, 64, 2, 2), (101, 37, 3, 2), (50, 50, 1, 1), (129, 257, 4, 4)])
def test_cyclic_import_export_roundtrip(m, n, dx, dy):
    """Upstream-style element-cyclic pieces <-> the dense operand of the GPU plans (matrix.hpp:8-11, util.hpp:135-164)."""
    from capital_amd.matrix import matrix, cyclic_import, cyclic_export
    a = np.random.default_rng(m + n).standard_normal((m, n))
    dense = matrix(n, m, 1, 1)
    for x in range(dx):
        for y in range(dy):
            P = matrix(n, m, dx, dy).from_numpy(orc.cyclic_local(a, x, y, dx, dy))
            cyclic_import(P, dense, x, y, dx, dy)
    assert np.array_equal(dense.to_numpy(), a)
    for x in range(dx):
        for y in range(dy):
            P = matrix(n, m, dx, dy)
            P.view().fill_(7.0)
            cyclic_export(dense, P, x, y, dx, dy)
            assert np.array_equal(P.to_numpy(), orc.cyclic_local(a, x, y, dx, dy))     # incl. the zero padding


def test_factor_from_cyclic_pieces_matches_reference_semantics():
    """d = 2 upstream layout: assemble the 4 pieces of distribute_symmetric, factor, split R back into pieces."""
    from capital_amd import cholinv
    from capital_amd.matrix import matrix, cyclic_import, cyclic_export
    n, d = 200, 2
    dense = matrix(n, n, 1, 1)
    for x in range(d):
        for y in range(d):
            P = matrix(n, n, d, d); P.distribute_symmetric(x, y, d, d, 0, True)
            cyclic_import(P, dense, x, y, d, d)
    a = orc.symmetric_global(n, True)
    assert np.array_equal(dense.to_numpy(), a)
    pack = cholinv.info(1, 1, -2, 'U'); cholinv.factor(dense, pack, None)
    R = cholinv.construct_R(pack)
    r_ref, _ = orc.cholinv(a, 1, 1, -2, 2, 2)
    for x in range(d):
        for y in range(d):
            P = matrix(n, n, d, d); cyclic_export(R, P, x, y, d, d)
            ref = orc.cyclic_local(r_ref, x, y, d, d)
            assert np.linalg.norm(P.to_numpy() - ref) <= 1e-13 * np.linalg.norm(r_ref)


def test_descriptor_ownership_and_pinned_staging_round_trip():
    """cap_desc_*: the allocating and the injection constructor (matrix.hpp:5-74), and host <-> HBM through the
    double-buffered pinned chunks: several chunks, ragged sizes, strided host images, pinned host memory (direct path)."""
    import ctypes as C
    from capital_amd import _lib
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(0)
    for (gx, gy, px, py) in [(300, 200, 1, 1), (1000, 9001, 2, 3), (2, 9_000_000, 1, 1), (4097, 4099, 1, 1)]:
        d = C.c_void_p()
        _lib.check(L.cap_desc_create(C.byref(d), gx, gy, px, py))
        lx, ly, ld = L.cap_desc_get(d, 2), L.cap_desc_get(d, 3), L.cap_desc_get(d, 4)
        assert (lx, ly) == (-(-gx // px), -(-gy // py)) and ld >= ly and L.cap_desc_get(d, 5) == 1
        ldh = ly + 3                                            # strided host image
        host = rng.standard_normal((lx, ldh))
        _lib.check(L.cap_desc_import_host(d, host.ctypes.data, ldh, s))
        back = np.full((lx, ldh), np.nan)
        _lib.check(L.cap_desc_export_host(d, back.ctypes.data, ldh, s))
        assert np.array_equal(back[:, :ly], host[:, :ly]) and np.isnan(back[:, ly:]).all()
        # the same device buffer seen through a view descriptor (injection constructor): not owned, same content
        v = C.c_void_p()
        _lib.check(L.cap_desc_create_view(C.byref(v), gx, gy, px, py, L.cap_desc_data(d), ld))
        assert L.cap_desc_get(v, 5) == 0 and L.cap_desc_data(v) == L.cap_desc_data(d)
        pinned = torch.empty(lx, ly, dtype=torch.float64).pin_memory()
        _lib.check(L.cap_desc_export_host(v, pinned.data_ptr(), ly, s))        # pinned host memory: direct engine copy
        assert np.array_equal(pinned.numpy(), host[:, :ly])
        L.cap_desc_destroy(v)                                    # must not free d's buffer
        _lib.check(L.cap_desc_export_host(d, back.ctypes.data, ldh, s))
        assert np.array_equal(back[:, :ly], host[:, :ly])
        L.cap_desc_destroy(d)


def test_global_host_import_export_of_both_descriptor_kinds():
    """cap_desc_import_host_global / export_host_global: every grid position of several grids picks exactly its own blocks (block-cyclic
    kind, cap_desc_create_bc) or elements (element-cyclic kind + cap_desc_set_position) out of a GLOBAL column-major host matrix and puts
    them back where they came from - bit-exact against NumPy index arithmetic, strided host images, ragged blocks, empty pieces, pieces
    of several pinned chunks (the 2-D packing runs on host threads while the previous chunk is on the PCIe link)."""
    import ctypes as C
    from capital_amd import _lib, dist_cholesky as dc
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(4)

    def device_piece(d):
        lx, ly, ld = L.cap_desc_get(d, 2), L.cap_desc_get(d, 3), L.cap_desc_get(d, 4)
        out = np.empty((lx, ly))
        _lib.check(L.cap_desc_export_host(d, out.ctypes.data, ly, s))          # the LOCAL export: the piece as it sits in HBM
        return out.T                                                           # [row, col]

    for (gy, gx, nb, Pr, Pc) in [(1000, 1000, 128, 2, 2), (2049, 777, 256, 2, 4), (250, 250, 128, 2, 4), (5000, 4200, 512, 1, 3), (130, 9000, 64, 3, 1)]:
        ldh = gy + 5
        host = np.zeros((gx, ldh)); host[:, :gy] = rng.standard_normal((gx, gy))        # [col][row], ld = ldh
        a = host[:, :gy].T                                                     # [row, col]
        back = np.zeros_like(host)
        for pr in range(Pr):
            for pc in range(Pc):
                d = C.c_void_p()
                _lib.check(L.cap_desc_create_bc(C.byref(d), gx, gy, nb, Pr, Pc, pr, pc, None, 0))
                rows = dc.global_index_2d(gy, nb, Pr, pr); cols = dc.global_index_2d(gx, nb, Pc, pc)
                assert (L.cap_desc_get(d, 3), L.cap_desc_get(d, 2)) == (rows.size, cols.size)
                _lib.check(L.cap_desc_import_host_global(d, host.ctypes.data, ldh, s))
                if rows.size and cols.size:
                    assert np.array_equal(device_piece(d), a[np.ix_(rows, cols)])
                _lib.check(L.cap_desc_export_host_global(d, back.ctypes.data, ldh, s))
                L.cap_desc_destroy(d)
        assert np.array_equal(back, host)                                       # every element written exactly where it came from, padding untouched
    # element-cyclic kind (upstream's layout, matrix.hpp:8-11)
    for (gy, gx, px, py) in [(1001, 777, 2, 2), (4096, 300, 1, 4), (513, 2050, 3, 2)]:
        host = rng.standard_normal((gx, gy)); a = host.T
        back = np.zeros_like(host)
        for x in range(px):
            for y in range(py):
                d = C.c_void_p()
                _lib.check(L.cap_desc_create(C.byref(d), gx, gy, px, py))
                assert L.cap_desc_import_host_global(d, host.ctypes.data, gy, s) == 1     # position unknown: refused
                _lib.check(L.cap_desc_set_position(d, x, y))
                _lib.check(L.cap_desc_import_host_global(d, host.ctypes.data, gy, s))
                want = orc.cyclic_local(a, x, y, px, py)
                nr, nc = len(range(y, gy, py)), len(range(x, gx, px))
                assert np.array_equal(device_piece(d)[:nr, :nc], want[:nr, :nc])
                _lib.check(L.cap_desc_export_host_global(d, back.ctypes.data, gy, s))
                L.cap_desc_destroy(d)
        assert np.array_equal(back, host)
    # a piece of several 64 MiB chunks: 9000 x 20000 on a 1 x 2 grid = 687 MiB per process
    gy, gx, nb = 9000, 20000, 512
    host = rng.standard_normal((gx, gy)); back = np.zeros_like(host)
    for pc in range(2):
        d = C.c_void_p()
        _lib.check(L.cap_desc_create_bc(C.byref(d), gx, gy, nb, 1, 2, 0, pc, None, 0))
        _lib.check(L.cap_desc_import_host_global(d, host.ctypes.data, gy, s))
        _lib.check(L.cap_desc_export_host_global(d, back.ctypes.data, gy, s))
        L.cap_desc_destroy(d)
    assert np.array_equal(back, host)


def test_matrix_numpy_round_trip_uses_the_descriptor():
    from capital_amd.matrix import matrix
    a = np.random.default_rng(1).standard_normal((777, 333))
    A = matrix(333, 777, 1, 1).from_numpy(a)
    assert np.array_equal(A.to_numpy(), a)
    assert np.array_equal(A.view().cpu().numpy(), a)
