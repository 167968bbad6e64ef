"""Static guard of the sweep kernel's performance contract, CPU-only (hipcc emits the gfx950 ISA of
gso_sweep2.hip): inside the ring loops of gso_sweep2_kernel<3> — the blocks that hold an LDS-DMA issue and a wait —

  * every `s_waitcnt vmcnt` is the COMPILE-TIME count of the ring (INFL - 2): one hipcc-placed `vmcnt(0)` in such a
    loop drains the DMA pipe on every step (what `settle()` is for);
  * there is no scratch access: the kernel spills (127 VGPRs at four waves per SIMD), but only in the per-row code
    around the loops;
  * the fused AXPY + Gram pass exists (DPP row shifts of `wave_sum_i32` inside ring loops) and the streamed mu rows are
    read as 8-byte elements (round 4: rows of doubles, no 4-byte planes).

A parity test would notice none of these; a 20 % slowdown would be the only symptom."""
import os
import re
import shutil
import subprocess

import pytest

import conftest as C

SRC = os.path.join(C.ROOT, "fplll_amd", "csrc", "gso_sweep2.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value",
         "-Wno-inline-asm", "--cuda-device-only"]
KERNEL = "_ZN5fphip2s217gso_sweep2_kernelILi3EEE"


def _hipcc():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


pytestmark = pytest.mark.skipif(_hipcc() is None, reason="needs hipcc")


@pytest.fixture(scope="module")
def blocks(tmp_path_factory):
    from fplll_amd import build
    asm = str(tmp_path_factory.mktemp("isa") / "sweep2.s")
    subprocess.check_call([_hipcc()] + FLAGS + build.PER_FILE_FLAGS.get("gso_sweep2.hip", []) + ["-S", "-o", asm, SRC],
                          stderr=subprocess.DEVNULL)
    src = open(asm).read()
    m = re.search(re.escape(KERNEL) + r"[A-Za-z0-9_]*:(.*?)\.Lfunc_end", src, re.S)
    assert m, "gso_sweep2_kernel<3> not found in the ISA"
    out, cur = [], None
    for line in m.group(1).split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", line):
            cur = []
            out.append(cur)
        elif cur is not None and line.startswith("\t") and not line.strip().startswith((".", ";")):
            cur.append(line.strip())
    return out


def sweep2_infl(nq):
    """Cfg<NQ>::INFL of gso_sweep2.h (LDS-DMA entries a wave keeps in flight), from the header's constants."""
    src = open(os.path.join(C.ROOT, "fplll_amd", "csrc", "gso_sweep2.h")).read()
    lds3 = int(re.search(r"#define FPHIP_S2_NQ3_LDS (\d+)", src).group(1))
    wave_lds = 13312 if nq == 4 else lds3 if nq == 3 else 9984
    npair = min(16, wave_lds // (2 * nq * 256))
    return 2 * (npair - 1)


def _ring(blocks):
    return [b for b in blocks if any("global_load_lds" in i for i in b) and any(i.startswith("s_waitcnt vmcnt") for i in b)]


def test_ring_loops_wait_with_the_compile_time_count_only(blocks):
    ring = _ring(blocks)
    assert len(ring) >= 200, len(ring)
    want = "s_waitcnt vmcnt(%d)" % (sweep2_infl(3) - 2)
    bad = [i for b in ring for i in b if i.startswith("s_waitcnt vmcnt") and i != want]
    assert not bad, ("a wait inside a ring loop is not the counted one", sorted(set(bad)))


def test_ring_loops_hold_no_scratch_access(blocks):
    ring = _ring(blocks)
    bad = [i for b in ring for i in b if i.startswith("scratch_")]
    assert not bad, bad[:5]
    total = sum(1 for b in blocks for i in b if i.startswith("scratch_"))
    assert total <= 80, "the per-row code spills more than it did (48 scratch instructions in round 4): %d" % total


def test_fused_pass_and_double_rows_are_in_the_loops(blocks):
    ring = _ring(blocks)
    fused = [b for b in ring if any("row_shr" in i or "row_bcast" in i for i in b)]
    assert len(fused) >= 12, "the AXPY fused into the Gram pass (wave_sum_i32's DPP scan) is gone from the ring loops"
    wide = [b for b in ring if any(i.startswith("ds_read_b64") or i.startswith("ds_read2_b64") or i.startswith("ds_read_b128") for i in b)]
    assert len(wide) >= 24, "the mu rows are no longer read as 8-byte elements"
