"""In-tree builds of the native libraries (no JIT cache: the .so files travel with the tree).

  libqpdo_amd.so  -- the product: C host driver + HIP kernels for gfx950 (hipcc)
  libqpdo_gen.so  -- synthetic problem generator (gcc, OpenMP); workload synthesis only
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB_PATH = os.path.join(_HERE, "libqpdo_amd.so")
GEN_PATH = os.path.join(_HERE, "libqpdo_gen.so")

HIP_SOURCES = ["qpdo_dev.hip", "qpdo_small.hip"]
C_SOURCES = ["qpdo_api.c"]
HEADERS = ["qpdo_dev.h", os.path.join(INCLUDE, "qpdo.h"), os.path.join(INCLUDE, "qpdo_amd_ext.h")] + \
    sorted(os.path.join("dev", f) for f in os.listdir(os.path.join(CSRC, "dev")) if f.endswith(".inc"))


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    for s in sources:
        s = s if os.path.isabs(s) else os.path.join(CSRC, s)
        if os.path.exists(s) and os.path.getmtime(s) > t:
            return True
    return False


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def build_gen(force=False):
    src = os.path.join(CSRC, "qpdo_gen.c")
    if force or _stale(GEN_PATH, [src]):
        subprocess.check_call(["gcc", "-O3", "-std=gnu11", "-fPIC", "-fopenmp", "-shared",
                               "-o", GEN_PATH, src, "-lm"])
    return GEN_PATH


def ensure_gen():
    if os.path.exists(GEN_PATH) and not _stale(GEN_PATH, [os.path.join(CSRC, "qpdo_gen.c")]):
        return GEN_PATH
    return build_gen()


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 for the kernels, plain C for the host driver."""
    if not (force or _stale(LIB_PATH, HIP_SOURCES + C_SOURCES + HEADERS)):
        return LIB_PATH
    cc = hipcc()
    if cc is None:
        raise RuntimeError("hipcc not found: libqpdo_amd.so cannot be built")
    objs = []
    common = ["-O3", "-fPIC", "-I", INCLUDE, "-I", CSRC]

    def hip_object(src):
        o = os.path.join(CSRC, src + ".o")
        # -amdgpu-mfma-vgpr-form: accumulators of the fp64 MFMA loops stay in VGPRs; the default heuristics put them in AGPRs inside
        # the loop and in VGPRs across its back edge (64 v_accvgpr moves and a drained matrix pipeline per 16 MFMAs in k_ldl_syrk)
        cmd = [cc, "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
               "-mllvm", "-amdgpu-mfma-vgpr-form", *common, "-c", os.path.join(CSRC, src), "-o", o]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        try:
            subprocess.check_call(cmd)
        except subprocess.CalledProcessError:
            # a compiler without that LLVM option: build without it (correct, the dense trailing update is ~7 % slower)
            i = cmd.index("-amdgpu-mfma-vgpr-form")
            del cmd[i - 1:i + 1]
            subprocess.check_call(cmd)
        return o
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(HIP_SOURCES)) as ex:          # the translation units side by side
        objs.extend(ex.map(hip_object, HIP_SOURCES))
    for s in C_SOURCES:
        o = os.path.join(CSRC, s + ".o")
        subprocess.check_call(["gcc", "-std=gnu11", "-ffp-contract=off", "-fopenmp", "-Wall", *common,
                               "-c", os.path.join(CSRC, s), "-o", o])
        objs.append(o)
    # the host driver is gcc/OpenMP code (parallel CSC -> CSR conversions in qpdo_setup): GNU OpenMP runtime
    gomp = subprocess.check_output(["gcc", "-print-file-name=libgomp.so"], text=True).strip()
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs, "-lm", "-lpthread",
                           "-L/opt/rocm/lib", "-lrccl", gomp])
    return LIB_PATH


def build_lib_testhooks(out_dir):
    """A TEST build of the product library with -DQPDO_TEST_HOOKS (fault injection into the chained triangular solves:
    QPDO_DENSE_CHAIN_INJECT).  The hooks are compiled out of libqpdo_amd.so; this variant goes to out_dir and is loaded through
    QPDO_AMD_LIB by the one test that needs it.  Only qpdo_dev.hip is recompiled; the other objects are the product's."""
    cc = hipcc()
    if cc is None:
        raise RuntimeError("hipcc not found")
    build_lib()
    os.makedirs(out_dir, exist_ok=True)
    o = os.path.join(out_dir, "qpdo_dev_testhooks.o")
    subprocess.check_call([cc, "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-DQPDO_TEST_HOOKS",
                           "-O3", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-c", os.path.join(CSRC, "qpdo_dev.hip"), "-o", o])
    so = os.path.join(out_dir, "libqpdo_amd_testhooks.so")
    gomp = subprocess.check_output(["gcc", "-print-file-name=libgomp.so"], text=True).strip()
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, o, os.path.join(CSRC, "qpdo_small.hip.o"),
                           os.path.join(CSRC, "qpdo_api.c.o"), "-lm", "-lpthread", "-L/opt/rocm/lib", "-lrccl", gomp])
    return so


def ensure_lib():
    """Path of the product library; builds it when the sources are newer.  Never falls back
    to anything else: a missing library is an error.  (QPDO_AMD_LIB: an explicitly named build of the same library,
    for A/B timing of two versions on one box; tools/ab_c4.py.)"""
    override = os.environ.get("QPDO_AMD_LIB")
    if override:
        if not os.path.exists(override):
            raise RuntimeError("QPDO_AMD_LIB=%s does not exist" % override)
        return override
    if os.path.exists(LIB_PATH) and not _stale(LIB_PATH, HIP_SOURCES + C_SOURCES + HEADERS):
        return LIB_PATH
    if hipcc() is None:
        if os.path.exists(LIB_PATH):
            return LIB_PATH
        raise RuntimeError("libqpdo_amd.so is missing and hipcc is unavailable; "
                           "run __graft_entry__.build() where ROCm is installed")
    return build_lib()


def build_all(force=False, verbose=False):
    return build_gen(force), build_lib(force, verbose)


def build_abi_driver(out_dir, sanitize=False):
    """Compiles tests/abi_driver.c -- a plain-C caller that includes only include/qpdo.h -- and links it against the
    product library.  sanitize=True: the host driver (qpdo_api.c, gcc) is rebuilt with -fsanitize=address,undefined and
    linked with the already compiled device objects into libqpdo_amd_asan.so inside out_dir (the product library in
    the tree is not touched); the driver is instrumented too.  Returns the path of the executable."""
    root = os.path.dirname(_HERE)
    src = os.path.join(root, "tests", "abi_driver.c")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "abi_driver_asan" if sanitize else "abi_driver")
    if not sanitize:
        lib = ensure_lib()
        subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-Werror", "-I", INCLUDE, src, "-o", exe,
                               "-L", os.path.dirname(lib), "-l:" + os.path.basename(lib), "-Wl,-rpath," + os.path.dirname(lib), "-lm"])
        return exe
    cc = hipcc()
    if cc is None:
        raise RuntimeError("hipcc not found")
    objs = [os.path.join(CSRC, s + ".o") for s in HIP_SOURCES]
    if not all(os.path.exists(o) for o in objs):
        build_lib(force=True)
    api = os.path.join(out_dir, "qpdo_api_asan.o")
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"]
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", *san, "-ffp-contract=off", "-fopenmp", "-fPIC", "-Wall", "-I", INCLUDE, "-I", CSRC,
                           "-c", os.path.join(CSRC, "qpdo_api.c"), "-o", api])
    so = os.path.join(out_dir, "libqpdo_amd_asan.so")
    rt = [subprocess.check_output(["gcc", "-print-file-name=" + n], text=True).strip() for n in ("libgomp.so", "libasan.so", "libubsan.so")]
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, api, *objs, "-lm", "-lpthread", "-L/opt/rocm/lib", "-lrccl", *rt])
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", *san, "-Wall", "-I", INCLUDE, src, "-o", exe, "-L", out_dir, "-lqpdo_amd_asan",
                           "-Wl,-rpath," + out_dir, "-lm"])
    return exe
