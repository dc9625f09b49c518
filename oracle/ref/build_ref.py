#!/usr/bin/env python3
"""Build the REAL reference (tbennun/capital) as a CPU oracle -> oracle/_ref/.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (capital_amd/) may
import, link or execute anything produced here.  Consumers: tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg.

What it does
------------
* compiles the reference's header-only C++14 library *from where it lies* in
  /root/reference, through three small drivers we wrote (oracle/ref/drv_*.cpp),
  against MPICH 3.3.2 + MKL 2021.4 found under /opt/conda (LP64 `libmkl_rt`);
* the upstream tree does not compile as shipped with GCC 11 (template-parameter
  shadowing, two `static_assert(0)`, an undeclared `T`, an undeclared
  `globalNumRows`, one missing `template` keyword).
  Fourteen one-line substitutions in six headers fix that; none touches
  arithmetic.  The substitutions are applied to a THROW-AWAY copy of the
  headers in a temp dir that is deleted afterwards - reference sources are
  never written into this repository;
* upstream ships no `mkl.h`; a declarations-only shim (oracle/ref/inc/mkl.h,
  written by us, prototypes of the 7 CBLAS/LAPACKE entry points the reference
  calls: blas/interface.hpp:54,74,92, lapack/interface.hpp:39,54,69,84) is used.

Outputs (git-ignored, NOT gpurun-ignored, so they travel to the GPU box):
  oracle/_ref/cholinv_ref   argv: N complete_inv split bcMult layout chunks policy [dump]
  oracle/_ref/cacqr_ref     argv: variant M N complete_inv split bcMult [dump]
  oracle/_ref/summa_ref     argv: op M N K c layout num_chunks alpha beta dump   (GEMM / TRMM / SYRK overloads of matmult::summa)
  oracle/_ref/{cholinv,cacqr,summa}_cap   the same three drivers linked with libcapital_amd_cblas.so in MKL's place (the reference
                            running on the product's operators; LD_LIBRARY_PATH = capital_amd/lib or tests/hipshim/_build/cblas)
  oracle/_ref/{cholinv,cacqr,summa}_engine   the same drivers with INTEGRATION.md section A (examples/engine_binding/*.inc) pasted over
                            upstream's double specialisations of blas::engine / lapack::engine: runs where host memory is device memory
                            (LD_LIBRARY_PATH = tests/hipshim/_build/engine, the CPU stand-in)
Run as: MKL_NUM_THREADS=1 /opt/conda/bin/mpiexec -n {1|8} oracle/_ref/cholinv_ref ...

The reference has no build system we can use (config.mk is an empty template,
Makefiles hard-code $HOME/capital) - this script is the recipe.
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("CAPITAL_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "oracle", "_ref")
CONDA = os.environ.get("CAPITAL_CONDA", "/opt/conda")


def _patch(path, subs):
    s = open(path).read()
    for a, b in subs:
        if a not in s:
            raise RuntimeError("patch anchor missing in %s: %r" % (path, a))
        s = s.replace(a, b)
    open(path, "w").write(s)


def _apply_compile_fixes(root):
    """The 14 compile-only substitutions."""
    T = ('template<typename ScalarType = double, typename DimensionType = int64_t, '
         'typename StructurePolicy = rect, typename OffloadPolicy = OffloadEachGemm>')
    _patch(os.path.join(root, 'src/matrix/matrix.h'), [
        (T, T.replace('ScalarType =', 'ScalarType_ =').replace('DimensionType =', 'DimensionType_ =')),
        ('  using ScalarType = ScalarType;', '  using ScalarType = ScalarType_;'),
        ('  using DimensionType = DimensionType;', '  using DimensionType = DimensionType_;'),
        ('static_assert(0,"not implemented"); return -1;', 'return -1;')])
    _patch(os.path.join(root, 'src/alg/cholesky/cholinv/cholinv.h'), [
        ('  template<typename ScalarType, typename DimensionType>\n  class info{',
         '  template<typename ScalarType_, typename DimensionType_>\n  class info{'),
        ('    using ScalarType = ScalarType;', '    using ScalarType = ScalarType_;'),
        ('    using DimensionType = DimensionType;', '    using DimensionType = DimensionType_;')])
    _patch(os.path.join(root, 'src/alg/qr/cacqr/cacqr.h'), [
        ('  template<typename ScalarType, typename DimensionType, typename CholeskyInversionType>\n  class info{',
         '  template<typename ScalarType_, typename DimensionType_, typename CholeskyInversionType>\n  class info{'),
        ('    using ScalarType = ScalarType;', '    using ScalarType = ScalarType_;'),
        ('    using DimensionType = DimensionType;', '    using DimensionType = DimensionType_;'),
        ('typename CholeskyInversionType::info<ScalarType,DimensionType> cholesky_inverse_args;',
         'typename CholeskyInversionType::template info<ScalarType,DimensionType> cholesky_inverse_args;')])
    _patch(os.path.join(root, 'src/alg/qr/cacqr/cacqr.hpp'), [
        ('{ static_assert(0,"not implemented"); }', '{ }')])
    _patch(os.path.join(root, 'src/matrix/structure.hpp'), [
        ('numElems*sizeof(T));', 'numElems*sizeof(ScalarType));')])
    _patch(os.path.join(root, 'src/util/util.hpp'), [
        ('  U globalX = CommInfo.x; U globalY = CommInfo.y; U index=0;',
         '  U globalX = CommInfo.x; U globalY = CommInfo.y; U index=0; '
         'U globalNumRows=Matrix.num_rows_global(); U globalNumColumns=Matrix.num_columns_global();')])


def _apply_engine_binding(root):
    """INTEGRATION.md section A made real: upstream's `double` specialisations of blas::engine / lapack::engine are cut out of the throw-away
    copy and examples/engine_binding/*.inc (the text a maintainer would paste) put in their place; the includes the binding needs go into
    src/util/shared.h, which every upstream header pulls in."""
    import re
    bind = os.path.join(REPO, "examples", "engine_binding")
    for rel, inc, names in (("src/blas/interface.hpp", "blas_interface_double.inc", ("_gemm", "_trmm", "_syrk")),
                            ("src/lapack/interface.hpp", "lapack_interface_double.inc", ("_potrf", "_trtri"))):
        path = os.path.join(root, rel)
        s = open(path).read()
        first = None
        for nm in names:
            m = re.search(r"template<>\s*\nvoid engine::%s\(double\*.*?\n\}\n" % nm, s, re.S)
            if not m:
                raise RuntimeError("engine binding: upstream's %s specialisation not found in %s" % (nm, rel))
            first = m.start() if first is None else min(first, m.start())
            s = s[:m.start()] + s[m.end():]
        s = s[:first] + open(os.path.join(bind, inc)).read() + "\n" + s[first:]
        open(path, "w").write(s)
    _patch(os.path.join(root, "src/util/shared.h"), [
        ('#include "mkl.h"', '#include "mkl.h"\n#include <stdexcept>\n#include <new>\n#include <hip/hip_runtime_api.h>\n#include "capital_amd.h"')])


def available():
    return (os.path.isdir(os.path.join(REF, "src", "alg")) and
            os.path.exists(os.path.join(CONDA, "lib", "libmkl_rt.so")) and
            os.path.exists(os.path.join(CONDA, "include", "mpi.h")))


def build(verbose=True):
    if not available():
        raise RuntimeError("reference tree / MKL / MPICH not present; cannot build oracle/_ref")
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="capital_ref_")
    try:
        root = os.path.join(tmp, "ref")
        # throw-away copy of headers only (src/ + test/), patched in place
        shutil.copytree(os.path.join(REF, "src"), os.path.join(root, "src"))
        shutil.copytree(os.path.join(REF, "test"), os.path.join(root, "test"))
        _apply_compile_fixes(root)
        inc = os.path.join(tmp, "inc")
        os.makedirs(inc)
        shutil.copy(os.path.join(HERE, "inc", "mkl.h"), inc)
        for h in ("mpi.h", "mpio.h", "mpicxx.h"):
            shutil.copy(os.path.join(CONDA, "include", h), inc)
        for drv, exe in (("drv_cholinv.cpp", "cholinv_ref"), ("drv_cacqr.cpp", "cacqr_ref"), ("drv_summa.cpp", "summa_ref")):
            cmd = ["g++", "-std=c++14", "-O2", "-fpermissive", "-w", "-DMPICH_SKIP_MPICXX",
                   "-I" + inc, "-I" + tmp, os.path.join(HERE, drv), "-o", os.path.join(OUT, exe),
                   "-L" + os.path.join(CONDA, "lib"), "-Wl,-rpath," + os.path.join(CONDA, "lib"),
                   "-lmpi", "-lmkl_rt", "-lpthread", "-lm", "-ldl"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            # the same driver with libcapital_amd_cblas.so (include/capital_amd_cblas.h) in MKL's place: the REAL reference, unmodified, with
            # every BLAS / LAPACK call served by the product's operators.  No rpath to either build of that library: LD_LIBRARY_PATH picks
            # capital_amd/lib (the GPU) or tests/hipshim/_build/cblas (the CPU stand-in); built only when the product library is
            cap = os.path.join(REPO, "capital_amd", "lib")
            if os.path.exists(os.path.join(cap, "libcapital_amd_cblas.so")):
                # (compiled against the PRODUCT's stand-in for mkl.h, include/for_upstream/mkl.h, in front of the oracle's own: what INTEGRATION.md section 0 tells a maintainer to do)
                cmd = ["-I" + os.path.join(REPO, "include", "for_upstream") if x == "-I" + inc else x for x in cmd]
                cmd = cmd[:cmd.index(os.path.join(HERE, drv))] + ["-I" + inc] + cmd[cmd.index(os.path.join(HERE, drv)):]      # (mpi.h still comes from the oracle's include copy)
                cmd = cmd[:cmd.index("-o")] + ["-o", os.path.join(OUT, exe.replace("_ref", "_cap")), "-L" + os.path.join(CONDA, "lib"),
                                                 "-Wl,-rpath," + os.path.join(CONDA, "lib"), "-lmpi", "-L" + cap, "-lcapital_amd_cblas",
                                                 "-Wl,--allow-shlib-undefined", "-lpthread", "-lm", "-ldl"]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
        # the engine specialisations themselves bound to the library (examples/engine_binding/*.inc pasted over
        # upstream's).  The matrices stay where upstream allocates them, so this build only RUNS where host memory is device memory - the CPU
        # stand-in (tests/test_reference_offload.py); on a GPU it needs section A's other half (matrix<> on hipMalloc).  What it proves:
        # the pasted text compiles against upstream's declarations and calls the operators with the right arguments.
        cap = os.path.join(REPO, "capital_amd", "lib")
        if os.path.exists(os.path.join(cap, "libcapital_amd_cblas.so")):
            _apply_engine_binding(root)
            for drv, exe in (("drv_cholinv.cpp", "cholinv_engine"), ("drv_cacqr.cpp", "cacqr_engine"), ("drv_summa.cpp", "summa_engine")):
                cmd = ["g++", "-std=c++14", "-O2", "-fpermissive", "-w", "-DMPICH_SKIP_MPICXX", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(REPO, "include", "for_upstream"), "-I" + inc, "-I" + tmp,
                       "-I/opt/rocm/include", "-I" + os.path.join(REPO, "include"), os.path.join(HERE, drv), "-o", os.path.join(OUT, exe),
                       "-L" + os.path.join(CONDA, "lib"), "-Wl,-rpath," + os.path.join(CONDA, "lib"), "-lmpi", "-L" + cap, "-lcapital_amd_cblas", "-lcapital_amd",
                       "-L/opt/rocm/lib", "-lamdhip64", "-Wl,--allow-shlib-undefined", "-lpthread", "-lm", "-ldl"]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return OUT


if __name__ == "__main__":
    build()
    print("built", OUT)
