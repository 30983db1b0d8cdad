"""Build the HIP shared library (and the pybind11 front end) in-tree for gfx950.

``python -m toast_amd.build`` or ``toast_amd.build.build()``.  hipcc cross-compiles
without a GPU; the resulting ``toast_amd/libtoast_hip.so`` travels to the GPU box with the
source tree.  -ffp-contract=off is REQUIRED: the pixel path must round every multiply and
add separately, like the reference's x86-64 build (see csrc/hpix_math.hpp).
"""

import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtoast_hip.so")

HIP_SOURCES = ["runtime.cpp", "arena.cpp", "vmm_slab.cpp", "kernels.hip", "otf_kernels.hip", "offset_prior.hip", "ground_filter.hip", "capi_host.cpp", "comm.cpp", "fft_filter.hip", "fft_fused.hip", "fft_reg.hip", "deterministic.hip", "pcg.hip", "packed_pointing.hip"]
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libtoast_hip can only be built with ROCm")
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _all_deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".hpp", ".inc"))]
    deps.append(os.path.join(HERE, "..", "include", "toast_hip.h"))
    return deps


def build_library(force=False, verbose=True):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not _newer(LIB, _all_deps()):
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        # TOAST_HIP_EXTRA_FLAGS: experimental builds (e.g. -DTOAST_FFT_PHASE_CLOCK, tools/exp_fft_phases.py)
        cmd = [_hipcc(), "-x", "hip", "-c", s, "-o", o] + HIPCC_FLAGS + os.environ.get("TOAST_HIP_EXTRA_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    link = [_hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return LIB


def build_pybind(force=False, verbose=True):
    import pybind11

    src = os.path.join(CSRC, "pybind_module.cpp")
    if not os.path.exists(src):
        return None
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(HERE, "_libtoast_hip" + ext)
    if not force and not _newer(out, [src, LIB, os.path.join(HERE, "..", "include", "toast_hip.h")]):
        return out
    cmd = [
        "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
        "-I" + sysconfig.get_paths()["include"], "-I" + pybind11.get_include(),
        "-I" + os.path.join(HERE, "..", "include"),
        src, "-o", out, "-L" + HERE, "-ltoast_hip", "-Wl,-rpath,$ORIGIN",
    ]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=True):
    lib = build_library(force=force, verbose=verbose)
    mod = build_pybind(force=force, verbose=verbose)
    return lib, mod


if __name__ == "__main__":
    build(force="--force" in sys.argv)
