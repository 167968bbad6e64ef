"""In-tree build of the native pieces (no JIT cache: the built .so files travel with the tree).

  fplll_amd/lib/libfplll_hip.so          HIP kernels + C ABI (include/fplll_hip.h), gfx950 only
  fplll_amd/lib/libfplll_hip_extenum.so  host C++ adapter for fplll::set_external_enumerator
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")

HIP_SOURCES = ["enum_kernel.hip", "enum_host.hip", "gso_kernel.hip", "gso_sweep2.hip", "lll_kernel.hip", "hlll_kernel.hip", "bkz_kernel.hip", "bkzs_kernel.hip", "bkzd_kernel.hip", "gso_host.hip"]
HIP_HEADERS = ["enum_device.h", "gso_device.h", "gso_wave.h", "gso_sweep2.h", "lll_wave.h", os.path.join(ROOT, "include", "fplll_hip.h")]
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17",
    "-ffp-contract=off",  # fplll's arithmetic is separate mul/add (nr/nr_FP_d.inl:178); no FMA
    "-fPIC", "-Wno-unused-value", "-Wno-inline-asm",
]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd, cwd=None):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=cwd)


def hipcc():
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        try:
            subprocess.check_output([c, "--version"], stderr=subprocess.STDOUT)
            return c
        except Exception:
            continue
    raise RuntimeError("hipcc not found")


def build_hip(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    out = os.path.join(LIBDIR, "libfplll_hip.so")
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]
    extra = []
    if os.environ.get("FPHIP_GSO_RING"):
        extra.append("-DFPHIP_GSO_RING=" + os.environ["FPHIP_GSO_RING"])
        force = True
    # one object per source (objects are scratch: fplll_amd/lib/obj is git-ignored), then link
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    relink = force or not os.path.exists(out)
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            _run([hipcc()] + HIPCC_FLAGS + extra + ["-c", "-o", obj, src])
            relink = True
    if relink:
        _run([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs)
    return out


def build_shim(force=False):
    out = os.path.join(LIBDIR, "libfplll_hip_extenum.so")
    src = os.path.join(CSRC, "extenum_shim.cpp")
    if force or _newer(out, [src, os.path.join(ROOT, "include", "fplll_hip.h")]):
        _run(["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-o", out, src, "-L" + LIBDIR,
              "-lfplll_hip", "-Wl,-rpath,$ORIGIN", "-pthread"])
    return out


def build_oracle():
    """Test infrastructure: the C restatement, and (only where /root/reference exists) the real
    reference into oracle/_ref/.  Building the checker is not using it."""
    odir = os.path.join(ROOT, "oracle")
    _run(["make", "-s", "port"], cwd=odir)
    if os.path.isdir("/root/reference/fplll"):
        _run(["make", "-s", "-j8", "ref"], cwd=odir)


def build_all(force=False):
    build_hip(force)
    build_shim(force)
    build_oracle()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
