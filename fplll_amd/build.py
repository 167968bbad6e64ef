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

HIP_SOURCES = ["enum_kernel.hip", "enum_walk.hip", "enum_deal.hip", "enum_host.hip", "gso_kernel.hip", "gso_sweep2.hip", "lll_kernel.hip", "lll_kernel_early.hip", "hlll_kernel.hip", "hh_blocked.hip", "hh_rows.hip", "hlll_x.hip", "lll_x.hip", "bkz_kernel.hip", "bkzs_kernel.hip", "gso_host.hip", "pruner_volume.hip", "pruner_search.hip", "gso_util_host.hip"]
HIP_HEADERS = ["dev_mem.h", "dev_cache.h", "trace.h", "pruner_tables.h", "pruner_engine.h", "enum_device.h", "enum_wave.h", "gso_device.h", "gso_wave.h", "gso_sweep2.h", "ftx.h", "lll_wave.h", "lll_stream.h", os.path.join(ROOT, "include", "fplll_hip.h")]
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17",
    "-ffp-contract=off",  # fplll's arithmetic is separate mul/add (nr/nr_FP_d.inl:178); no FMA
    "-fPIC", "-Wno-unused-value", "-Wno-inline-asm",
]


# Per-file flags.  enum_kernel.hip: its hot loops hold wave-uniform branches only; without this
# option the AMDGPU backend still structurises them (flag registers, exit codes, a copy of every
# live register per iteration) because the surrounding function has lane-masked branches — the
# option makes StructurizeCFG leave regions of uniform branches alone (see the kernel's header).
PER_FILE_FLAGS = {
    # (-disable-lifetime-markers: with lifetime markers every `break` out of a loop body that declares
    #  locals goes through the front end's cleanup block — ONE block shared with the loop latch, a
    #  "continue" flag and a phi per live value: 4 scalar + 2 branch instructions per iteration of the
    #  walk loops, which are bound by the scalar / branch issue port)
    "enum_kernel.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1", "-Xclang", "-disable-lifetime-markers"],
    "enum_walk.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1", "-Xclang", "-disable-lifetime-markers"],
    "bkzs_kernel.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1", "-Xclang", "-disable-lifetime-markers"],
    # the one-wavefront-per-lattice reduction kernels: their loops still hold lane-masked branches, so
    # the option only spares the regions that are uniform already — measured +4 % on the batched LLL
    # (100.5 -> 104.6 lattices/s at d = 120, batch 1024), outputs unchanged (parity tests)
    "lll_kernel.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1"],
    "lll_kernel_early.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1"],
    "bkz_kernel.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1"],
    "hlll_kernel.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1"],
    "gso_kernel.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1"],
}


# sources that include another source
EXTRA_DEPS = {"lll_kernel_early.hip": ["lll_kernel.hip"]}


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
    jobs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        more = [os.path.join(CSRC, x) for x in EXTRA_DEPS.get(os.path.basename(src), [])]
        if force or _newer(obj, [src, os.path.abspath(__file__)] + hdrs + more):  # (the flags live in this file)
            jobs.append([hipcc()] + HIPCC_FLAGS + extra + PER_FILE_FLAGS.get(os.path.basename(src), []) +
                        ["-c", "-o", obj, src])
            relink = True
    if jobs:  # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
            list(ex.map(_run, jobs))
    if relink:
        _run([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", "-o", out] + objs)
    return out


def build_shim(force=False):
    out = os.path.join(LIBDIR, "libfplll_hip_extenum.so")
    src = os.path.join(CSRC, "extenum_shim.cpp")
    if force or _newer(out, [src, os.path.join(ROOT, "include", "fplll_hip.h")]):
        _run(["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-o", out, src, "-L" + LIBDIR,
              "-lfplll_hip", "-Wl,-rpath,$ORIGIN", "-pthread"])
    return out


REF_HEADERS = "/root/reference"  # fplll's source tree (headers only are used)
CONDA = "/opt/conda"            # gmp.h / mpfr.h


def build_gso_dropin(force=False):
    """libfplll_hip_gso.so: MatGSOHip + the LLLReduction::lll / babai specialisations and MatHouseholderHip + the
    HLLLReduction::hlll specialisation (csrc/dropin/), host C++
    against fplll's own headers — what a maintainer builds inside fplll's tree.  Where the headers
    are absent (the GPU box) the prebuilt library that travelled with the tree is used."""
    out = os.path.join(LIBDIR, "libfplll_hip_gso.so")
    src = os.path.join(CSRC, "dropin", "matgso_hip.cpp")
    src2 = os.path.join(CSRC, "dropin", "mathouseholder_hip.cpp")
    hdr = os.path.join(CSRC, "dropin", "matgso_hip.h")
    hdr2 = os.path.join(CSRC, "dropin", "mathouseholder_hip.h")
    if not os.path.isdir(os.path.join(REF_HEADERS, "fplll")):
        return out if os.path.exists(out) else None
    if force or _newer(out, [src, src2, hdr, hdr2, os.path.join(ROOT, "include", "fplll_hip.h")]):
        # fplll_config.h is generated by fplll's configure; the values of a default build
        cfg = os.path.join(LIBDIR, "obj", "fplll_cfg", "fplll")
        os.makedirs(cfg, exist_ok=True)
        with open(os.path.join(cfg, "fplll_config.h"), "w") as f:
            f.write("#ifndef FPLLL_CONFIG__H\n#define FPLLL_CONFIG__H\n#define FPLLL_MAJOR_VERSION 5\n"
                    "#define FPLLL_MINOR_VERSION 5\n#define FPLLL_MICRO_VERSION 0\n#define FPLLL_VERSION 5.5.0\n"
                    "#define FPLLL_VERSION_INFO 9:0:0\n#define FPLLL_MAX_ENUM_DIM 256\n"
                    "#define FPLLL_WITH_RECURSIVE_ENUM 1\n#define FPLLL_MAX_PARALLEL_ENUM_DIM 120\n"
                    "#define HAVE_LIBGMP 1\n#endif\n")
        inc = os.path.dirname(cfg)
        # (enum/enumerate_base.h includes "../fplll_config.h": one level up as well)
        import shutil
        shutil.copy(os.path.join(cfg, "fplll_config.h"), os.path.join(inc, "fplll_config.h"))
        _run(["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-w", "-I" + inc, "-I" + cfg,
              "-I" + REF_HEADERS, "-I" + os.path.join(REF_HEADERS, "fplll"), "-I" + os.path.join(CONDA, "include"),
              "-o", out, src, src2, "-L" + LIBDIR, "-lfplll_hip", "-Wl,-rpath,$ORIGIN", "-ldl", "-pthread"])
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
    build_gso_dropin(force)
    build_oracle()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
