"""CPU-only: the C-ABI library builds/loads without a GPU, exports every symbol include/capital_amd.h declares,
the ctypes table covers the header, and the product path refuses to run on CPU buffers (no fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "capital_amd.h")
SO = os.path.join(ROOT, "capital_amd", "lib", "libcapital_amd.so")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(cap_[A-Za-z0-9_]+)\s*\(", src))
    names -= {"cap_allgather_fn", "cap_bcast_fn", "cap_allreduce_fn"}       # function-pointer typedefs
    return sorted(names)


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(SO):
        from capital_amd import build
        build.build(verbose=False)          # hipcc cross-compiles for gfx950 without a GPU
    return SO


def test_header_symbols_are_exported(built):
    lib = ctypes.CDLL(built)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, "declared in include/capital_amd.h but not exported: %s" % missing
    assert len(_declared()) >= 50


def test_ctypes_table_matches_header(built):
    from capital_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    _lib.lib()                              # loads and type-checks every entry


def test_status_strings_and_pure_helpers(built):
    from capital_amd import _lib
    L = _lib.lib()
    assert L.cap_status_string(0) == b"ok" and L.cap_status_string(3) == b"matrix is not positive definite"
    assert L.cap_dpotrf_work_size(1024) > 0 and L.cap_dtrsm_work_size(0, 128, 64) >= 128 * 128 + 128 * 64
    assert L.cap_bc_owner(11, 8) == 3 and L.cap_bc_local_block(11, 8) == 1


def test_block_cyclic_and_grid_maps_property(built):
    """Pure index maps of the C ABI against the reference's definitions, on random shapes (hypothesis): the 1 x P
    block-column-cyclic layout (every block column has one owner and a dense local slot, local column counts add up, ragged
    last block included) and the rank -> (x, y, z) maps of topo::square / topo::rect (topology.h:44-50,75-83) - bijective,
    and equal to the Python mirror the tests use."""
    import ctypes as C
    from hypothesis import given, settings, strategies as st
    from capital_amd import _lib, topo
    L = _lib.lib()

    @settings(max_examples=200, deadline=None)
    @given(st.integers(1, 70000), st.sampled_from([64, 128, 256, 512, 1024]), st.integers(1, 16))
    def bc(n, nb, P):
        nblk = (n + nb - 1) // nb
        slots = set()
        for J in range(nblk):
            o, lb = L.cap_bc_owner(J, P), L.cap_bc_local_block(J, P)
            assert o == J % P and lb == J // P
            slots.add((o, lb))
        assert len(slots) == nblk
        cols = [L.cap_bc_num_local_cols(n, nb, P, p) for p in range(P)]
        assert sum(cols) == n and all(c >= 0 for c in cols)
        for p in range(P):                     # blocks J = p, p + P, ...; the last global block may be ragged
            exp = sum(min(nb, n - J * nb) for J in range(p, nblk, P))
            assert cols[p] == exp

    @settings(max_examples=100, deadline=None)
    @given(st.integers(1, 4), st.integers(1, 6), st.booleans())
    def grid(c, d, square):
        size = c * d * d if square else c * c * d
        seen = set()
        for rank in range(size):
            dd, x, y, z = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            assert L.cap_topo_coords(0 if square else 1, rank, size, c, C.byref(dd), C.byref(x), C.byref(y), C.byref(z)) == 0
            ref = (topo.square_coords if square else topo.rect_coords)(rank, size, c)
            assert (dd.value, x.value, y.value, z.value) == (ref["d"], ref["x"], ref["y"], ref["z"])
            assert 0 <= z.value < c and 0 <= y.value < d and 0 <= x.value < (d if square else c)
            seen.add((x.value, y.value, z.value))
        assert len(seen) == size

    bc(); grid()


def test_hot_kernels_use_no_scratch(built):
    """The compiler's own resource remarks, saved by the build: hot kernels must keep their accumulators in registers."""
    from capital_amd import build as b
    if not b.kernel_resources():
        b.build(force=True, verbose=False)
    res = b.check_no_scratch()
    hot = [k for k in res if any(n in k for n in b.NO_SCRATCH)]
    assert len(hot) >= 6
    for k in hot:
        assert int(res[k]["ScratchSize"]) == 0 and int(res[k]["VGPRs Spill"]) == 0, k


def test_qrapply_counted_requests_match_the_machine_code():
    """qrapply256_kernel waits for 'K tile u + 1 has landed' with a counted vmcnt; the count assumes, per half step, [nq + 2 LDS-DMA requests, then
    the 8 stores of a finished block column] and nothing else.  The build checks the kernel's machine code against that; so does this test."""
    from capital_amd import build as b
    assert b.check_qrapply_requests() == 32          # 16 K steps x 2 column parities


def test_no_cpu_fallback():
    """CPU tensors are rejected before any native call; there is no host path to fall back to."""
    import torch
    from capital_amd import _lib, blas
    a = torch.ones(4, 4, dtype=torch.float64)
    pack = blas.ArgPack_gemm(blas.Order.AblasColumnMajor, 0, 0, 1.0, 0.0)
    with pytest.raises(_lib.CapitalError):
        blas.engine._gemm(a, a, a, 4, 4, 4, 4, 4, 4, pack)
    with pytest.raises(_lib.CapitalError):
        blas.ArgPack_gemm  # noqa: B018
        blas.engine._gemm(a, a, a, 4, 4, 4, 4, 4, 4, blas.ArgPack_gemm(blas.Order.AblasRowMajor, 0, 0, 1.0, 0.0))


def test_reference_enum_values_and_topology_maps():
    from capital_amd import blas, lapack, topo
    # blas/engine.h:23-52, lapack/engine.h:23-52
    assert (blas.Transpose.AblasTrans, blas.Side.AblasRight, blas.UpLo.AblasUpper, blas.Diag.AblasUnit) == (1, 1, 1, 1)
    assert (blas.Method.AblasSyrk, lapack.Method.AlapackOrgqr) == (0x10, 0x11)
    # topology.h:81-83 (layout 0): rank = z + c*x + c*d*y on the 2x2x2 grid
    seen = set()
    for rank in range(8):
        t = topo.square_coords(rank, 8, 2)
        assert (t["c"], t["d"]) == (2, 2) and rank == t["z"] + 2 * t["x"] + 4 * t["y"]
        seen.add((t["x"], t["y"], t["z"]))
    assert len(seen) == 8
    r = topo.rect_coords(5, 8, 1)            # the 1D CholeskyQR grid: c = 1, d = 8, rows cyclic over y
    assert (r["c"], r["d"], r["x"], r["y"], r["z"]) == (1, 8, 0, 5, 0)


DRIVERS = ("cholinv_driver", "summa_driver", "cacqr_driver", "integration_snippets")


def _compile_c(built, name, out):
    """gcc -std=c99 (C, not C++) of examples/<name>.c against include/capital_amd.h, linked with the library"""
    import subprocess
    cmd = ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=199309L", "-Wall", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
           "-Werror=int-conversion", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", name + ".c"), "-L" + os.path.dirname(built), "-lcapital_amd", "-L/opt/rocm/lib",
           "-lamdhip64", "-lm", "-Wl,-rpath," + os.path.dirname(built), "-Wl,-rpath,/opt/rocm/lib", "-o", str(out)]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.parametrize("name", DRIVERS)
def test_plain_c_host_compiles_against_the_header(built, tmp_path, name):
    """north_star: "host code stays C" - C (not C++) translation units include the header and link the library: the three bench
    drivers (the C forms of bench/cholesky/cholinv.cpp, bench/matmult/summa_gemm.cpp, bench/qr/cacqr.cpp) and
    examples/integration_snippets.c, which holds every call INTEGRATION.md shows - wrong arity, a char where an enum is expected or a
    wrong pointer type is a compile error here, so the document cannot drift from the header again."""
    import shutil
    if not shutil.which("gcc") or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("gcc / ROCm headers not available")
    exe = tmp_path / (name + ".bin")
    r = _compile_c(built, name, exe)
    assert r.returncode == 0, r.stderr
    assert exe.exists()


def test_release_library_reads_no_experiment_switch_from_the_environment(built):
    """Round 5 review: ~40 CAP_* environment switches were live in the release library - a stray variable in a caller's environment silently
    changed the product's code path.  They are read through CAP_ENV now, which is getenv only in experiment builds (csrc/common.h:
    CAP_EXPERIMENTS, false in the tree) and folds to a null pointer otherwise: the names must not even be IN the release binary, and no
    plain getenv of a CAP_ name may be left in the sources.  (libcapital_amd_cblas.so keeps CAPCB_REPORT / CAPCB_DEVICE: a report and the
    device choice of an MPI program that has no line of its own to make it - not code-path switches.)"""
    import glob
    csrc = os.path.join(ROOT, "capital_amd", "csrc")
    common = open(os.path.join(csrc, "common.h")).read()
    assert "constexpr bool CAP_EXPERIMENTS = false;" in common, "the tree must hold a release configuration"
    names = set()
    for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        src = open(f).read()
        assert not re.search(r'(?<![A-Za-z_])getenv\("CAP_', src.replace("CAP_ENV", "")), "plain getenv of a CAP_ switch in " + f
        names.update(re.findall(r'CAP_ENV\("(CAP_[A-Z0-9_]+)"\)', src))
    assert len(names) >= 30, names
    blob = open(built, "rb").read()
    left = sorted(n for n in names if n.encode() + b"\0" in blob)
    assert not left, "experiment switches present in the release library: %s" % left


def test_integration_md_snippets_are_the_compiled_ones():
    """Every cap_* call line inside INTEGRATION.md's C / C++ code blocks of section B appears verbatim (modulo whitespace) in
    examples/integration_snippets.c - the file the test above compiles."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = re.sub(r"\s+", "", open(os.path.join(ROOT, "examples", "integration_snippets.c")).read())
    sec = md[md.index("## B."):md.index("## C.")]
    calls = []
    for block in re.findall(r"```cpp\n(.*?)```", sec, flags=re.S):
        for stmt in re.findall(r"\bcap_[a-z0-9_]+\s*\([^;]*?\)\s*;", block, flags=re.S):
            calls.append(stmt)
    assert len(calls) >= 40, len(calls)
    missing = [c for c in calls if re.sub(r"\s+", "", c) not in src]
    assert not missing, "INTEGRATION.md shows calls that are not in the compiled snippet file: %s" % missing


@pytest.mark.gpu
@pytest.mark.parametrize("name,argv,expect", [
    ("cholinv_driver", ("2048", "-1", "1", "-3", "1", "1"), "residual"),
    ("cholinv_driver", ("1024", "1", "1", "-2", "1", "1"), "residual"),
    ("summa_driver", ("768", "512", "640", "1", "0", "2", "2", "1"), "trmm"),
    ("cacqr_driver", ("2", "16384", "128", "1", "1", "1", "1", "0", "0", "0", "0", "1", "1"), "orthogonality"),
    ("cacqr_driver", ("1", "4096", "256", "1", "1", "1", "1", "0", "0", "0", "0", "1", "1"), "orthogonality"),
])
def test_plain_c_drivers_run_on_the_gpu(built, tmp_path, name, argv, expect):
    """The C drivers (no torch, no Python in the process) run on the GPU and pass their own validation blocks: the residual of
    test/cholesky/validate.hpp:33-46, the three summa::invoke overloads against the local operators, the CholeskyQR residual and
    orthogonality of test/qr/validate.hpp:24-51."""
    import subprocess
    exe = tmp_path / (name + ".bin")
    r = _compile_c(built, name, exe)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe)] + list(argv), capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert expect in run.stdout, run.stdout[-2000:]
    if name == "cholinv_driver":
        res = float(re.search(r"= ([0-9.eE+-]+)\s*$", run.stdout.strip().splitlines()[-1]).group(1))
        assert res < 1e-14, run.stdout


def test_2d_block_cyclic_index_maps(built):
    """Pure index helpers of the Pr x Pc layout (csrc/dist2d.hip, csrc/gemm.hip): local extents partition the matrix, and the
    staircase row count of the update kernel equals a brute-force count over local row tiles."""
    from capital_amd import _lib, dist_cholesky as dc
    L = _lib.lib()
    for (n, nb, Pr, Pc) in [(4096, 512, 2, 4), (1000, 128, 2, 2), (2049, 256, 2, 4), (65536, 512, 2, 4), (1536, 128, 4, 4), (777, 128, 1, 3)]:
        assert sum(L.cap_bc2d_local_extent(n, nb, Pr, Pc, pr, 0, 0) for pr in range(Pr)) == n
        assert sum(L.cap_bc2d_local_extent(n, nb, Pr, Pc, 0, pc, 1) for pc in range(Pc)) == n
        for pr in range(Pr):
            assert L.cap_bc2d_local_extent(n, nb, Pr, Pc, pr, 0, 0) == dc.global_index_2d(n, nb, Pr, pr).size
        for pc in range(Pc):
            assert L.cap_bc2d_local_extent(n, nb, Pr, Pc, 0, pc, 1) == dc.global_index_2d(n, nb, Pc, pc).size
    for nbT in (1, 2, 4):
        for Pr in (1, 2, 4):
            for pr in range(Pr):
                for J0 in (0, 1, 5, 6):
                    rlb0 = len([I for I in range(pr, J0, Pr)])            # local row blocks with I < J0
                    nloc = 7
                    gti = []                                             # global tile index (relative to J0) of every local row tile
                    for b in range(nloc):
                        I = pr + Pr * (rlb0 + b)
                        gti += [(I - J0) * nbT + t for t in range(nbT)]
                    for X in range(-2, (nloc * Pr + 2) * nbT):
                        want = sum(1 for g in gti if g <= X) if X >= 0 else 0
                        got = L.cap_bc2d_rows_le(X, nbT, J0, Pr, pr, rlb0)
                        if X < (pr + Pr * (rlb0 + nloc - 1) - J0 + 1) * nbT:     # beyond the local extent the kernel clamps to its tile count
                            assert got == want, (X, nbT, J0, Pr, pr, rlb0, got, want)


def test_redistribution_message_counts_property(built):
    """cap_redist_message_elems (pure, no GPU): the all-to-all of the element-cyclic <-> block-cyclic redistribution moves every element of the
    matrix exactly once per direction-0 pass (the replicas share the supply: destination t takes from layer t mod c only) and once PER LAYER on
    the way back, and what a rank sends to a peer is what a brute-force walk over the two index maps finds (matrix.hpp:8-11)."""
    import numpy as np
    from capital_amd import _lib
    L = _lib.lib()
    for (n, nb, P, c, Pr) in [(64, 16, 8, 2, 1), (64, 16, 8, 2, 2), (50, 8, 4, 1, 2), (37, 5, 9, 1, 3), (96, 32, 8, 2, 1), (20, 128, 4, 1, 1), (33, 4, 1, 1, 1)]:
        d = int(round((P / c) ** 0.5)); Pc = P // Pr
        tot0 = tot1 = 0
        for s in range(P):
            z, x, y = s % c, (s % (d * c)) // c, s // (d * c)
            for t in range(P):
                pr, pc = t // Pc, t % Pc
                rows = [g for g in range(y, n, d) if (g // nb) % Pr == pr]
                cols = [g for g in range(x, n, d) if (g // nb) % Pc == pc]
                want = len(rows) * len(cols)
                got0 = L.cap_redist_message_elems(n, nb, P, c, Pr, s, t, 0)
                got1 = L.cap_redist_message_elems(n, nb, P, c, Pr, t, s, 1)
                assert got0 == (want if z == t % c else 0), (n, nb, P, c, Pr, s, t)
                assert got1 == want
                tot0 += got0; tot1 += got1
        assert tot0 == n * n and tot1 == c * n * n
    assert L.cap_redist_message_elems(64, 16, 6, 1, 1, 0, 0, 0) == -1          # 6 ranks are no d x d x c grid


def test_block_cyclic_descriptor_extents(built):
    """The block-cyclic kind of cap_desc (cap_desc_create_bc; the injection constructor needs no GPU): valid local rows x columns
    per grid position agree with cap_bc2d_local_extent (what cap_dist2d_* allocate), add up to the global dimensions over the
    grid, the fields are readable, and a wrong position / leading dimension is refused."""
    import ctypes as C
    from capital_amd import _lib
    L = _lib.lib()
    fake = C.c_void_p(0x1000)                       # never dereferenced: descriptors of caller-owned buffers do no device work
    for (gx, gy, nb, Pr, Pc) in [(2048, 2048, 128, 2, 4), (1000, 1000, 128, 2, 2), (2049, 777, 256, 1, 8), (250, 250, 128, 2, 4),
                                 (65536, 65536, 512, 2, 4), (5, 9, 4, 3, 2)]:
        tot_r = [0] * Pc; tot_c = [0] * Pr
        for pr in range(Pr):
            for pc in range(Pc):
                d = C.c_void_p()
                assert L.cap_desc_create_bc(C.byref(d), gx, gy, nb, Pr, Pc, pr, pc, fake, max(gy, 1)) == 0
                lr, lc = L.cap_desc_get(d, 3), L.cap_desc_get(d, 2)
                assert lr == L.cap_bc2d_local_extent(gy, nb, Pr, Pc, pr, pc, 0) and lc == L.cap_bc2d_local_extent(gx, nb, Pr, Pc, pr, pc, 1)
                assert [L.cap_desc_get(d, f) for f in (0, 1, 5, 6, 7, 9, 10, 11, 12)] == [gx, gy, 0, Pc, Pr, 1, nb, pc, pr]
                tot_r[pc] += lr; tot_c[pr] += lc
                L.cap_desc_destroy(d)
        assert all(t == gy for t in tot_r) and all(t == gx for t in tot_c)
    d = C.c_void_p()
    assert L.cap_desc_create_bc(C.byref(d), 100, 100, 16, 2, 2, 2, 0, fake, 100) == 1        # pr out of range
    assert L.cap_desc_create_bc(C.byref(d), 100, 100, 16, 2, 2, 0, 0, fake, 10) == 1         # ld below the local rows
    assert L.cap_desc_create_bc(C.byref(d), 100, 100, 0, 2, 2, 0, 0, fake, 100) == 1
    # element-cyclic descriptors learn their position through cap_desc_set_position (needed by the global import / export)
    assert L.cap_desc_create_view(C.byref(d), 100, 100, 2, 2, fake, 50) == 0
    assert L.cap_desc_get(d, 9) == 0 and L.cap_desc_get(d, 11) == -1
    assert L.cap_desc_import_host_global(d, fake, 100, None) == 1                             # position unknown: refused
    assert L.cap_desc_set_position(d, 1, 0) == 0 and (L.cap_desc_get(d, 11), L.cap_desc_get(d, 12)) == (1, 0)
    assert L.cap_desc_set_position(d, 2, 0) == 1
    L.cap_desc_destroy(d)
