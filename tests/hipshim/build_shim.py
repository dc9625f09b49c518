"""TEST INFRASTRUCTURE: links the product's own object files against the recording HIP / RCCL stand-in (hipshim.cpp) instead of
libamdhip64 / librccl -> tests/hipshim/_build/libcapital_amd_shim.so (git-ignored).  No GPU, no HIP runtime in the process."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OBJ = os.path.join(ROOT, "capital_amd", "lib", "obj")
OUT = os.path.join(HERE, "_build")
SHIM = os.path.join(OUT, "libhipshim.so")
LIB = os.path.join(OUT, "libcapital_amd_shim.so")


def build():
    objs = sorted(glob.glob(os.path.join(OBJ, "*.o")))
    if not objs:
        raise RuntimeError("no object files under capital_amd/lib/obj - build the library first (python -m capital_amd.build)")
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, "hipshim.cpp")
    cpu = os.path.join(HERE, "kernels_cpu.cpp")          # CPU models of the kernels (compute mode), decoding the library's own argument structs
    csrc = os.path.join(ROOT, "capital_amd", "csrc")
    newest = max(os.path.getmtime(f) for f in objs + [src, cpu, os.path.join(csrc, "kargs.h"), os.path.join(csrc, "gemm_index.h"), os.path.abspath(__file__)])
    if os.path.exists(LIB) and os.path.exists(SHIM) and min(os.path.getmtime(LIB), os.path.getmtime(SHIM)) > newest:
        return LIB, SHIM
    cpu_o = os.path.join(OUT, "kernels_cpu.o")
    subprocess.check_call(["g++", "-std=c++17", "-O3", "-march=native", "-fopenmp", "-fPIC", "-Wall", "-Wno-unused-function", "-I" + csrc, "-c", cpu, "-o", cpu_o])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-fopenmp", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", SHIM, src, cpu_o])
    subprocess.check_call(["g++", "-shared", "-fPIC", "-o", LIB] + objs + ["-L" + OUT, "-lhipshim", "-Wl,-rpath," + OUT, "-Wl,--no-undefined", "-ldl", "-lpthread"])
    return LIB, SHIM


def build_cblas():
    """the CBLAS / LAPACKE offload library (include/capital_amd_cblas.h) over the stand-in: the product's own object file of it, linked
    against libcapital_amd_shim.so instead of libcapital_amd.so + libamdhip64 -> _build/cblas/libcapital_amd_cblas.so (same soname as the
    product's: a program linked with -lcapital_amd_cblas picks one or the other through LD_LIBRARY_PATH)"""
    import sys
    sys.path.insert(0, ROOT)
    from capital_amd import build as pb
    pb.build_cblas(verbose=False)
    lib, shim = build()
    out = os.path.join(OUT, "cblas")
    os.makedirs(out, exist_ok=True)
    dst = os.path.join(out, "libcapital_amd_cblas.so")
    if not os.path.exists(dst) or os.path.getmtime(dst) < max(os.path.getmtime(f) for f in (pb.CBLAS_OBJ, lib, shim)):
        subprocess.check_call(["g++", "-shared", "-fPIC", "-o", dst, pb.CBLAS_OBJ, "-L" + OUT, "-lcapital_amd_shim", "-lhipshim", "-Wl,-rpath," + OUT, "-Wl,--no-undefined"])
    return dst


def build_engine_dir():
    """a directory in which the PRODUCT's library names resolve to the stand-in builds (symbolic links: libcapital_amd.so -> the shim build,
    libamdhip64.so.7 -> libhipshim.so, libcapital_amd_cblas.so -> the build over the stand-in): a program linked against the product
    (oracle/_ref/*_engine: INTEGRATION.md section A pasted into the reference) runs on the CPU with LD_LIBRARY_PATH pointing here"""
    cb = build_cblas()
    d = os.path.join(OUT, "engine")
    os.makedirs(d, exist_ok=True)
    for name, target in (("libcapital_amd.so", LIB), ("libamdhip64.so.7", SHIM), ("libcapital_amd_cblas.so", cb)):
        link = os.path.join(d, name)
        if os.path.islink(link) or os.path.exists(link):
            os.unlink(link)
        os.symlink(target, link)
    return d


if __name__ == "__main__":
    print(build())
    print(build_cblas())
    print(build_engine_dir())
