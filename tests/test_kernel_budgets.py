"""Static guard of the register / scratch budgets the measured occupancies rest on (DESIGN.md sections 3, 4c, 4f, 5):
the code objects of the built library (fplll_amd/lib/obj/*.o, gfx950) carry them as metadata.  A kernel that slips
over one of these lines still computes the right thing — and loses a resident wave per SIMD, which no parity test
notices.  CPU-only: llvm-objcopy / clang-offload-bundler / llvm-readelf on the objects build() left behind."""
import os
import re
import subprocess

import pytest

import conftest as C

LLVM = "/opt/rocm/lib/llvm/bin"
TOOLS = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
pytestmark = pytest.mark.skipif(not all(os.path.exists(t) for t in TOOLS), reason="needs the ROCm LLVM tools")


def _kernels(src, tmp):
    """{mangled kernel name: (registers = VGPRs + AGPRs, scratch bytes per lane)} of one translation unit."""
    from fplll_amd import build
    build.build_hip()
    obj = os.path.join(C.ROOT, "fplll_amd", "lib", "obj", src + ".o")
    assert os.path.exists(obj), obj
    fat, co = os.path.join(tmp, src + ".fat"), os.path.join(tmp, src + ".co")
    subprocess.check_call([TOOLS[0], "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(tmp, "unused.o")])
    subprocess.check_call([TOOLS[1], "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.check_output([TOOLS[2], "--notes", co]).decode()
    out = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = (int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)),
                     int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)))
    return out


def _pick(ks, pattern):
    hit = {k: v for k, v in ks.items() if re.search(pattern, k)}
    assert hit, pattern
    return hit


# registers per lane for N resident waves per SIMD (512 per lane, allocated in eights)
WAVES = {8: 64, 5: 96, 4: 128, 3: 168, 2: 256}


def test_enumeration_walk_kernels(tmp_path):
    ks = _kernels("enum_kernel.hip", str(tmp_path))
    for k, (regs, scratch) in _pick(ks, "enum_phase_kernel").items():
        assert regs <= WAVES[8] and scratch == 0, (k, regs, scratch)  # 8 waves per SIMD, nothing in scratch memory
    for k, (regs, scratch) in _pick(ks, "enum_bfs_kernel|task_(pack|unpack|key)_kernel").items():
        assert scratch == 0, (k, scratch)


def test_lll_kernels_keep_their_waves(tmp_path):
    ks = _kernels("lll_kernel.hip", str(tmp_path))
    # the plain kernels (the batched LLL's throughput: 2048 lattices = two waves per SIMD at 120 dimensions)
    lim = {1: WAVES[3], 2: WAVES[2], 3: WAVES[2], 4: WAVES[2]}
    for nq, top in lim.items():
        (k, (regs, _)), = _pick(ks, r"lll_kernelILi%dELb0E" % nq).items()
        assert regs <= top, (k, regs, top)
    # ... and the extended ones (early reduction, u) live in their own translation unit
    assert not _pick(ks, "lll_kernel").keys() & _pick(_kernels("lll_kernel_early.hip", str(tmp_path)), "lll_kernel").keys()


def test_bkz_kernels_keep_their_waves(tmp_path):
    ks = _kernels("bkz_kernel.hip", str(tmp_path))
    for nq, top in {1: WAVES[3], 2: WAVES[2]}.items():  # config 2 (120 dimensions): two waves per SIMD
        (k, (regs, _)), = _pick(ks, r"bkz_kernelILi%dE" % nq).items()
        assert regs <= top, (k, regs, top)
    ks = _kernels("bkzs_kernel.hip", str(tmp_path))
    (k, (regs, _)), = _pick(ks, r"11bkzs_kernelILi1E").items()
    assert regs <= WAVES[2], (k, regs)  # big batches of lattices up to 64 columns: two waves per SIMD (397 BKZ-40/s)


def test_hlll_and_sweep_kernels_keep_their_waves(tmp_path):
    ks = _kernels("hlll_kernel.hip", str(tmp_path))
    for nq, top in {1: WAVES[5], 2: WAVES[4], 3: WAVES[3], 4: WAVES[2]}.items():
        (k, (regs, scratch)), = _pick(ks, r"hlll_kernelILi%dE" % nq).items()
        assert regs <= top and scratch == 0, (k, regs, top, scratch)
    ks = _kernels("gso_sweep2.hip", str(tmp_path))
    for nq, top in {1: WAVES[4], 2: WAVES[4], 3: WAVES[4], 4: WAVES[3]}.items():
        (k, (regs, _)), = _pick(ks, r"gso_sweep2_kernelILi%dE" % nq).items()
        assert regs <= top, (k, regs, top)
