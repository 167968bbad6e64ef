"""Static guard of the second-generation walk kernel's performance contract (enum_walk.hip): its
two hot loops — the EXPAND chain (all children of a node in one 64-lane test) and the STEP loop (the next sibling by
index) — hold wave-uniform branches only, no scratch traffic and no register-copy storms, stay within the static
instruction budgets the per-node PMC figures correspond to (profiles/r06_enum_walk2_pmc_*.txt: 26 VALU + 20 SALU + 8
branch + 5.5 LDS per node), and the kernel keeps 8 waves per SIMD.  The LDS unit is shared by the four SIMDs of a CU:
every ds_bpermute in these loops was measured to cost throughput, so their number is pinned as well.
CPU-only: hipcc emits the optimised IR / ISA for gfx950, `opt` prints the uniformity analysis."""
import os
import re
import shutil
import subprocess

import pytest

import conftest as C
from test_isa_uniform_loops import FLAGS, OPT, _hipcc, _innermost_loops, _kernel_body

SRC = os.path.join(C.ROOT, "fplll_amd", "csrc", "enum_walk.hip")
WALK = "_ZN5fphip16enum_walk_kernelILb0ELb0E"  # <MU_LDS = false, DUAL = false>: the big launches

pytestmark = pytest.mark.skipif(_hipcc() is None or not os.path.exists(OPT), reason="needs hipcc and opt")


@pytest.fixture(scope="module")
def artefacts(tmp_path_factory):
    from fplll_amd import build
    d = tmp_path_factory.mktemp("isa_walk2")
    ll, asm = str(d / "walk.ll"), str(d / "walk.s")
    per_file = build.PER_FILE_FLAGS["enum_walk.hip"]
    assert "-structurizecfg-skip-uniform-regions=1" in per_file and "-disable-lifetime-markers" in per_file
    front = [f for i, f in enumerate(per_file) if f != "-mllvm" and (i == 0 or per_file[i - 1] != "-mllvm")]
    subprocess.check_call([_hipcc()] + FLAGS + front + ["-S", "-emit-llvm", "-o", ll, SRC], stderr=subprocess.DEVNULL)
    subprocess.check_call([_hipcc()] + FLAGS + per_file + ["-S", "-o", asm, SRC], stderr=subprocess.DEVNULL)
    uni = subprocess.run([OPT, "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-passes=print<uniformity>",
                          "-disable-output", ll], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, check=True)
    return uni.stderr.decode(), open(asm).read()


def test_walk2_kernels_have_no_loop_with_a_divergent_exit(artefacts):
    uni, _ = artefacts
    seen = 0
    for p in uni.split("UniformityInfo for function ")[1:]:
        name = p.split("'")[1]
        if "enum_walk_kernel" not in name:
            continue
        seen += 1
        cycles = [l for l in p.split("\n") if l.strip().startswith("depth=") and len(l.split(")")[-1].split()) >= 2]
        assert not cycles, "%s: loops with a divergent exit: %s" % (name, cycles[:2])
    assert seen == 4


def test_walk2_hot_loops_within_their_budgets(artefacts):
    _, asm = artefacts
    body = _kernel_body(asm, WALK)
    found = {}
    for seg in _innermost_loops(body):
        # the EXPAND chain is the innermost loop with the vector roundto; the STEP loop is the readlane-and-compare
        # loop right behind it (its body is the code that follows the loop)
        if any(s.startswith("v_rndne_f64") for s in seg) and any(s.startswith("s_bcnt1_i32_b64") for s in seg):
            key = "expand"
        elif len(seg) <= 12 and any(s.startswith("v_readlane_b32") for s in seg) and any("s_bfe_u32" in s for s in seg):
            key = "step_check"
        else:
            continue
        found[key] = dict(valu=sum(s.startswith("v_") for s in seg),
                          salu=sum(s.startswith("s_") and not s.startswith(("s_nop", "s_waitcnt", "s_cbranch", "s_branch"))
                                   for s in seg),
                          lds=sum(s.startswith("ds_") for s in seg),
                          mov=sum(s.startswith("v_mov") for s in seg),
                          scratch=sum(s.startswith("scratch_") for s in seg),
                          execs=sum(bool(re.match(r"s_\w+\s+exec\b", s)) or "saveexec" in s for s in seg),
                          rl=sum(s.startswith(("v_readlane", "v_readfirstlane")) for s in seg))
    assert set(found) == {"expand", "step_check"}, found
    e = found["expand"]
    # (static counts of ALL blocks of the loop: the tie-rounding and global-stack blocks included)
    # (one v_readlane: the reload of a spilled scalar in the global-stack block)
    assert e["execs"] == 0 and e["scratch"] == 0 and e["rl"] <= 1, e
    assert e["valu"] <= 42 and e["salu"] <= 24 and e["mov"] <= 4 and e["lds"] <= 6, e
    s = found["step_check"]
    assert s["execs"] == 0 and s["scratch"] == 0 and s["valu"] <= 2 and s["salu"] <= 6 and s["lds"] == 0, s
    m = re.search(re.escape(WALK) + r"[^\n]*\n(?:.*\n)*?\s*\.vgpr_count:\s+(\d+)", asm[asm.index(".amdgpu_metadata"):])
    assert m and int(m.group(1)) <= 64, m and m.group(1)


def test_walk2_scratch_is_confined_to_the_slow_paths(artefacts):
    """64 VGPRs (8 waves per SIMD) with at most 16 bytes of scratch per lane, none of it touched by the hot loops."""
    _, asm = artefacts
    meta = asm[asm.index(".amdgpu_metadata"):]
    seen = 0
    for m in re.finditer(r"\.name:\s+(_ZN5fphip16enum_walk_kernel\S+)", meta):
        blk = meta[max(0, meta.rfind("- .agpr_count", 0, m.start())):meta.find("- .agpr_count", m.end())]
        seg = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
        vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
        assert seg and vg, m.group(1)
        assert int(seg.group(1)) <= 16 and int(vg.group(1)) <= 64, (m.group(1), seg.group(1), vg.group(1))
        seen += 1
    assert seen == 4
