"""In-tree build of the native pieces (explicit hipcc / g++; no JIT cache).

  eva_amd/lib/libeva_hip.so   — HIP kernels + C-ABI (hipcc --offload-arch=gfx950)
  eva_amd/_eva*.so            — host module (pybind11, g++), built when its sources exist
"""
import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
HIP_LIB = os.path.join(LIBDIR, "libeva_hip.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build the gfx950 backend)")


HIP_UNITS = ("runtime.hip", "elementwise.hip", "ewprogram.hip", "keyswitch.hip", "rotate.hip", "windows.hip", "shard.hip", "client.hip", "scheduler.hip")
HIP_HEADERS = tuple(sorted(f for f in os.listdir(CSRC) if f.endswith(".h")))  # every header of csrc/: a new one must not be missed (r6)


def build_hip(force=False, verbose=False):
    """The translation units of libeva_hip.so are compiled concurrently and linked into one shared
    library (runtime and scheduler hold no device code)."""
    units = [os.path.join(CSRC, f) for f in HIP_UNITS]
    srcs = units + [os.path.join(CSRC, f) for f in HIP_HEADERS]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "eva_hip.h"))
    if not force and not _newer(HIP_LIB, srcs):
        return HIP_LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    procs, objs = [], []
    for u in units:
        o = os.path.join(objdir, os.path.basename(u)[:-4] + ".o")
        objs.append(o)
        cmd = [_hipcc()] + flags + ["-c", u, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return HIP_LIB


def host_module_path():
    ext = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    return os.path.join(HERE, "_eva" + ext)


def build_host(force=False, verbose=False):
    hdir = os.path.join(HERE, "host")
    main = os.path.join(hdir, "module.cpp")
    if not os.path.exists(main):
        return None
    srcs = [os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith((".cpp", ".h"))]
    srcs += [os.path.join(CSRC, "hostmath.h"), os.path.join(os.path.dirname(HERE), "include", "eva_hip.h")]
    out = host_module_path()
    if not force and not _newer(out, srcs):
        return out
    import pybind11
    cpps = [s for s in srcs if s.endswith(".cpp")]
    # EVA_HOST_CXXFLAGS: extra flags for this one build (scripts/sanitize_cpu.sh: -fsanitize=address,undefined)
    # -ffp-contract=off: the FP64 encoder must round exactly like the device encoder (which is
    # compiled without FMA contraction too), so both give the same plaintext bit for bit
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-march=x86-64-v3", "-ffp-contract=off",
           "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"],
           "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", CSRC, "-I", hdir,
           "-o", out] + os.environ.get("EVA_HOST_CXXFLAGS", "").split() + cpps + ["-L", LIBDIR, "-leva_hip", "-Wl,-rpath,$ORIGIN/lib", "-lpthread", "-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_host(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
