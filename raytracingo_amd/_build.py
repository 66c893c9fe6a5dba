"""Build the native parts of raytracingo_amd in-tree (the .so files travel to the GPU box with the snapshot).

  librtgo_hip.so   csrc/rtgo_capi.hip + rtgo_device.h   hipcc --offload-arch=gfx950   (the product: C ABI + kernels)
  librtgo_host.so  host/*.cpp                            g++ (+ libamdhip64, librccl)  (Scene/Shape/Renderer mirror, multi-GPU driver)
  rtgo_engine      host/main.cpp                         g++                           (headless CLI of engine/main.cpp)
"""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: one IEEE rounding per operation, the same statement of the arithmetic as the oracle
# -fno-slp-vectorize: hipcc otherwise packs neighbouring scalar f32 mul/add into v_pk_*_f32; on gfx950 those issue at half
#   rate and cost v_mov's to pair their operands (and registers: the 5-waves variant spills with them) -- measured +18 %
# -amdgpu-atomic-optimizer-strategy=None: the optimizer rewrites the work queue's lane-0 atomicAdd into a wave-aggregated
#   one whose result is redistributed (and so waited for) on the spot; that turns the prefetched pull of the NEXT strip
#   into a 1-10 us stall per strip
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-slp-vectorize",
             "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall"]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
# the multi-GPU driver (host/multigpu.cpp) makes HIP runtime calls (streams, events, buffers) and RCCL calls from plain C++
HOST_ROCM_FLAGS = ["-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include")]
HOST_ROCM_LIBS = ["-L" + os.path.join(ROCM, "lib"), "-lamdhip64", "-lrccl", "-Wl,-rpath," + os.path.join(ROCM, "lib")]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


class _build_lock:
    """one builder at a time (N ranks of a multi-GPU launch import the package together); results appear by atomic rename"""

    def __enter__(self):
        import fcntl
        self.f = open(os.path.join(PKG, ".build.lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def _run_to(out, cmd_before_out, cmd_after_out, verbose):
    tmp = "%s.tmp.%d" % (out, os.getpid())
    cmd = cmd_before_out + ["-o", tmp] + cmd_after_out
    if verbose:
        print(" ".join(cmd_before_out + ["-o", out] + cmd_after_out))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def build_hip(force=False, verbose=False):
    src = os.path.join(PKG, "csrc", "rtgo_capi.hip")
    csrc = os.path.join(PKG, "csrc")
    deps = [src, os.path.join(ROOT, "include", "rtgo.h")] + [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".h", ".inc"))]
    out = os.path.join(PKG, "librtgo_hip.so")
    if force or _stale(out, deps):
        with _build_lock():
            if force or _stale(out, deps):
                _run_to(out, [HIPCC] + HIP_FLAGS, [src], verbose)
    return out


def host_sources():
    d = os.path.join(PKG, "host")
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".cpp") and f != "main.cpp")


def build_host(force=False, verbose=False):
    d = os.path.join(PKG, "host")
    if not os.path.isdir(d):
        return None
    srcs = host_sources()
    hdrs = [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")] + [os.path.join(ROOT, "include", "rtgo.h"),
                                                                              os.path.join(ROOT, "include", "rtgo_host.h")]
    out = os.path.join(PKG, "librtgo_host.so")
    if srcs and (force or _stale(out, srcs + hdrs)):
        with _build_lock():
            if force or _stale(out, srcs + hdrs):
                # (only gpu_transport.cpp includes HIP / RCCL headers; the define is harmless for the others)
                _run_to(out, ["g++"] + HOST_FLAGS + HOST_ROCM_FLAGS + ["-shared", "-I" + os.path.join(ROOT, "include"), "-I" + d],
                        srcs + ["-L" + PKG, "-lrtgo_hip", "-Wl,-rpath,$ORIGIN"] + HOST_ROCM_LIBS, verbose)
    exe = os.path.join(PKG, "rtgo_engine")
    main = os.path.join(d, "main.cpp")
    if os.path.exists(main) and (force or _stale(exe, [main, out] + hdrs)):
        with _build_lock():
            if force or _stale(exe, [main, out] + hdrs):
                _run_to(exe, ["g++"] + HOST_FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + d],
                        [main, "-L" + PKG, "-lrtgo_host", "-lrtgo_hip", "-Wl,-rpath,$ORIGIN"] + HOST_ROCM_LIBS, verbose)
    return out


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_host(force, verbose)
