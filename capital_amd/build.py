"""Build libcapital_amd.so (HIP kernels + C ABI + C++ host schedule) for gfx950, in-tree.

    python -m capital_amd.build        # or: from capital_amd.build import build; build()

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with gpurun.
"""
import glob
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcapital_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage", "-I" + os.path.join(os.path.dirname(HERE), "include")]
# Kernels of the hot path that must live in registers: a silent scratch allocation (seen once: accumulators captured by a
# lambda) costs 5-7x and nothing else reports it.  Checked after every build against the compiler's own resource remarks.
NO_SCRATCH = ("dgemm_tn_dma_kernel", "gram256_kernel", "qrapply256_kernel", "bf16_tn_kernel", "bf16_tn_v2_kernel", "bf16_tn3_kernel", "bf16_tn3w_kernel", "bf16_tn3x_kernel", "leaf_cholinv_kernel",
              "panel64_solve_update_kernel", "chain64_coop_kernel")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(HERE), "include", "capital_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(HERE), "include", "capital_amd.h")]
        if (not force and os.path.exists(obj) and
                all(os.path.getmtime(obj) > os.path.getmtime(d) for d in [src] + hdrs)):
            continue
        cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for src, obj, p in procs:
        _, err = p.communicate()
        remarks = [l for l in err.splitlines() if "kernel-resource-usage" in l]
        other = [l for l in err.splitlines() if "kernel-resource-usage" not in l and not re.match(r"^\s*\d*\s*\|", l)
                 and "remarks generated" not in l and not l.startswith("In file included from")]   # (drop the source excerpt / include trail clang prints with every remark)
        if other:
            print("\n".join(other), file=sys.stderr, flush=True)
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on " + src)
        with open(obj + ".resources.txt", "w") as f:
            f.write("\n".join(remarks) + "\n")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl",
                                                                                      "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    check_no_scratch()
    check_qrapply_requests()
    return LIB


CBLAS_LIB = os.path.join(LIBDIR, "libcapital_amd_cblas.so")
CBLAS_SRC = os.path.join(CSRC, "cblas", "cblas_offload.cpp")
CBLAS_OBJ = os.path.join(LIBDIR, "obj_cblas", "cblas_offload.o")


def build_cblas(force=False, verbose=True):
    """libcapital_amd_cblas.so (include/capital_amd_cblas.h): the seven CBLAS / LAPACKE symbols the reference imports, on host pointers,
    served by libcapital_amd.so's operators through staging buffers.  Its own library: a process that also holds a CPU BLAS must not find
    cblas_dgemm in libcapital_amd.so.  Host code only (g++; the HIP runtime API header, no device code)."""
    build(force=False, verbose=verbose)
    inc = os.path.join(os.path.dirname(HERE), "include")
    deps = [CBLAS_SRC, os.path.join(inc, "capital_amd_cblas.h"), os.path.join(inc, "capital_amd.h"), LIB]
    if not force and os.path.exists(CBLAS_LIB) and all(os.path.getmtime(CBLAS_LIB) > os.path.getmtime(d) for d in deps):
        return CBLAS_LIB
    os.makedirs(os.path.dirname(CBLAS_OBJ), exist_ok=True)
    for cmd in (["g++", "-std=c++17", "-O2", "-fPIC", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + inc, "-c", CBLAS_SRC, "-o", CBLAS_OBJ],
                ["g++", "-shared", "-fPIC", "-o", CBLAS_LIB, CBLAS_OBJ, "-L" + LIBDIR, "-lcapital_amd", "-L/opt/rocm/lib", "-lamdhip64",
                 "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"]):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return CBLAS_LIB


REPLAY_LIB = os.path.join(LIBDIR, "libcap_replay.so")
REPLAY_SRC = os.path.join(CSRC, "replay", "replay_comm.hip")


def build_replay(force=False, verbose=True):
    """libcap_replay.so (csrc/replay/replay_comm.hip): the replay communicator of tools/replay.py / bench.py --replay-rank - a measurement
    harness ABOVE the C ABI (it only calls cap_comm_create_callbacks), kept out of libcapital_amd.so."""
    build(force=False, verbose=verbose)
    deps = [REPLAY_SRC, os.path.join(os.path.dirname(HERE), "include", "capital_amd.h"), LIB]
    if not force and os.path.exists(REPLAY_LIB) and all(os.path.getmtime(REPLAY_LIB) > os.path.getmtime(d) for d in deps):
        return REPLAY_LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", REPLAY_LIB, REPLAY_SRC, "-L" + LIBDIR, "-lcapital_amd",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return REPLAY_LIB


def kernel_resources():
    """{kernel symbol: {"VGPRs": n, "ScratchSize": bytes per lane, "VGPRs Spill": n, ...}} from the last compile of every source."""
    out, cur = {}, None
    for f in sorted(glob.glob(os.path.join(LIBDIR, "obj", "*.resources.txt"))):
        for line in open(f):
            body = line.split("remark:", 1)[-1].split("[-Rpass")[0].strip()
            if body.startswith("Function Name:"):
                cur = out.setdefault(body.split(":", 1)[1].strip(), {})
            elif cur is not None and ":" in body:
                k, v = body.rsplit(":", 1)
                cur[k.split("[")[0].strip()] = v.strip()
    return out


def device_text_md5(source="gemm.hip"):
    """md5 of the machine code (.text of the gfx950 code object) the last compile of `source` produced - the identity of its KERNELS:
    host-only edits of the source leave it unchanged (the bundle around it carries a path-dependent compilation-unit id, so the object
    file itself does not compare).  None when the object or the LLVM tools are missing."""
    import hashlib
    import tempfile
    obj = os.path.join(LIBDIR, "obj", source + ".o")
    tools = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(obj) or not os.path.exists(os.path.join(tools, "clang-offload-bundler")):
        return None
    try:
        with tempfile.TemporaryDirectory() as d:
            fat, co, txt = (os.path.join(d, x) for x in ("fat.bin", "code.co", "text.bin"))
            subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], stderr=subprocess.DEVNULL)
            subprocess.check_call([os.path.join(tools, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.DEVNULL)
            subprocess.check_call([os.path.join(tools, "llvm-objcopy"), "-O", "binary", "--only-section=.text", co, txt], stderr=subprocess.DEVNULL)
            return hashlib.md5(open(txt, "rb").read()).hexdigest()
    except Exception:
        return None


def check_no_scratch():
    res = kernel_resources()
    bad = [(k, v) for k, v in res.items() if any(n in k for n in NO_SCRATCH)
           and (int(v.get("ScratchSize", 0)) != 0 or int(v.get("VGPRs Spill", 0)) != 0)]
    if bad:
        raise RuntimeError("hot kernels use scratch memory: " + ", ".join("%s (%s B/lane)" % (k, v.get("ScratchSize")) for k, v in bad))
    return res


def check_qrapply_requests():
    """qrapply256_kernel (csrc/cqr_kernels.hip) certifies "K tile u + 1 has landed" with a COUNTED s_waitcnt vmcnt: the count is the number of
    younger requests by construction - per half step [nq pieces of R^-1 + 2 rows of Q as LDS-DMA, THEN the 8 stores of a finished block column].
    The order and the counts are the compiler's to keep (sched_group_barrier only asks); this check reads the kernel's machine code and fails
    when a region between two barriers does not hold exactly the expected requests in that order.  Returns the number of regions checked."""
    src = os.path.join(CSRC, "cqr_kernels.hip")
    asm = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(os.path.dirname(HERE), "include"), "-S",
                          "--cuda-device-only", "-o", "-", src], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    m = re.search(r"^(_ZN\S*qrapply256_kernelILi0E\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel", asm, re.S | re.M)
    if not m:
        raise RuntimeError("qrapply256_kernel<0> not found in the device assembly")
    regions, cur = [], []
    for line in m.group(2).splitlines():
        t = line.strip()
        if t.startswith("s_barrier"):
            regions.append(cur); cur = []
        elif t.startswith("buffer_load_dwordx4") and " lds" in t:
            cur.append("L")
        elif t.startswith("buffer_store_dwordx2"):
            cur.append("S")
        elif t.startswith(("buffer_", "global_", "flat_", "scratch_")):
            cur.append("?")                      # any other vector memory instruction would break the count
    regions.append(cur)
    # straight-line steps: prologue (A0 B0 A1 B1 A2 | B2 A3), then for parity 0 and parity 1 the 16 steps of a row tile (second half of every step)
    nq = lambda kt: 4 - ((kt & 15) >> 2)
    expect = []
    for par in (0, 1):
        for kt in range(16):
            expect.append("L" * (nq(kt + 3) + 2) + ("S" * 8 if (kt & 1) == par else ""))
    got = ["".join(r) for r in regions if r]
    body = list(got)
    for pro in ("L" * 14, "L" * 6):               # the prologue's two request groups (A0 B0 A1 B1 A2 | B2 A3)
        if pro not in body:
            raise RuntimeError("qrapply256_kernel: prologue requests not found: %s" % got)
        body.remove(pro)
    if any("?" in g for g in got):
        raise RuntimeError("qrapply256_kernel: unexpected vector memory instruction next to the counted requests: %s" % got)
    if sorted(body) != sorted(expect) or any(("S" in g and "L" in g[g.index("S"):]) for g in body):
        raise RuntimeError("qrapply256_kernel: requests per half step are not [LDS-DMA ..., stores ...] as the counted vmcnt assumes:\n got %s\n expected %s" % (body, expect))
    return len(body)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
    print("built", build_cblas(force="--force" in sys.argv))
    print("built", build_replay(force="--force" in sys.argv))
