"""Static guard of the enumeration walk's performance contract: the two hot loops of
enum_phase_kernel hold wave-uniform branches only.  One lane-masked branch inside them (or a merged
join that makes LLVM's uniformity analysis call a loop exit divergent) brings back the structuriser's
exit codes, flag registers and per-iteration register copies — a 1.7x slowdown that no parity test
would notice.  CPU-only: hipcc emits the optimised IR / ISA for gfx950, `opt` prints the uniformity
analysis.  (The method and the other helpers: tests/perf/isa_uniformity.sh, isa_bbstat.py.)"""
import os
import re
import shutil
import subprocess

import pytest

import conftest as C

SRC = os.path.join(C.ROOT, "fplll_amd", "csrc", "enum_kernel.hip")
OPT = "/opt/rocm/lib/llvm/bin/opt"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value",
         "-Wno-inline-asm", "--cuda-device-only"]
WALK = "_ZN5fphip17enum_phase_kernelILb0ELb0ELb0"  # <MU_LDS=false, SUBS=false, DUAL=false>: the big launches


def _hipcc():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


pytestmark = pytest.mark.skipif(_hipcc() is None or not os.path.exists(OPT), reason="needs hipcc and opt")


@pytest.fixture(scope="module")
def artefacts(tmp_path_factory):
    from fplll_amd import build
    d = tmp_path_factory.mktemp("isa")
    ll, asm = str(d / "enum.ll"), str(d / "enum.s")
    per_file = build.PER_FILE_FLAGS.get("enum_kernel.hip", [])
    # (the IR the uniformity analysis reads is built with the front-end options of the real build; the
    #  -mllvm ones only act in the code generator)
    front = [f for i, f in enumerate(per_file) if f != "-mllvm" and (i == 0 or per_file[i - 1] != "-mllvm")]
    subprocess.check_call([_hipcc()] + FLAGS + front + ["-S", "-emit-llvm", "-o", ll, SRC], stderr=subprocess.DEVNULL)
    subprocess.check_call([_hipcc()] + FLAGS + per_file + ["-S", "-o", asm, SRC], stderr=subprocess.DEVNULL)
    uni = subprocess.run([OPT, "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-passes=print<uniformity>",
                          "-disable-output", ll], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, check=True)
    return uni.stderr.decode(), open(asm).read()


def test_build_uses_the_skip_uniform_regions_flag():
    from fplll_amd import build
    assert "-structurizecfg-skip-uniform-regions=1" in build.PER_FILE_FLAGS["enum_kernel.hip"]
    # without it every `break` of the walk loops goes through the front end's cleanup block: a merged
    # latch with a "continue" flag (4 scalar + 2 branch instructions per iteration)
    assert "-disable-lifetime-markers" in build.PER_FILE_FLAGS["enum_kernel.hip"]
    assert "-structurizecfg-skip-uniform-regions=1" in build.PER_FILE_FLAGS["bkzs_kernel.hip"]


def test_walk_kernels_have_no_loop_with_a_divergent_exit(artefacts):
    uni, _ = artefacts
    parts = uni.split("UniformityInfo for function ")
    seen = 0
    for p in parts[1:]:
        name = p.split("'")[1]
        # the variants without sub-solution reporting (its ring hand-off has lane-masked branches)
        if not re.match(r"_ZN5fphip17enum_phase_kernelILb[01]ELb0ELb[01]", name):
            continue
        seen += 1
        # (a cycle of one block is the lane-strided copy of the mu rows into LDS, not a walk loop)
        cycles = [l for l in p.split("\n") if l.strip().startswith("depth=") and len(l.split(")")[-1].split()) >= 2]
        assert not cycles, "%s: loops with a divergent exit: %s" % (name, cycles[:2])
    assert seen == 4


def _kernel_body(asm, prefix):
    lines = asm.split("\n")
    st = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l)
    en = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    return lines[st:en]


def _innermost_loops(body):
    """Blocks of every innermost loop, from the compiler's own annotations: the header carries
    'This Inner Loop Header', its blocks 'in Loop: Header=BB<n>' (exit blocks that merely sit between
    them in the layout belong to the parent loop and are not counted)."""
    blocks, cur = [], None
    for l in body:
        m = re.match(r"^(\.LBB(\d+_\d+)):(.*)$", l)
        if m or l.startswith("; %bb."):
            cur = {"name": m.group(2) if m else None, "comment": (m.group(3) if m else l), "lines": []}
            blocks.append(cur)
        elif cur is not None:
            s = l.strip()
            if s.startswith(";") and ("Loop" in s):
                cur["comment"] += " " + s
            elif s:
                cur["lines"].append(s)
    loops = []
    for i, b in enumerate(blocks):
        if "This Inner Loop Header" in b["comment"] and b["name"]:
            members = [b] + [c for c in blocks if re.search(r"in Loop: Header=BB%s\b" % b["name"], c["comment"])]
            loops.append([s for c in members for s in c["lines"]])
    return loops


def test_hot_loops_are_free_of_exec_masking_and_copy_storms(artefacts):
    """The CHILD chain (recognised by its v_rndne_f64: roundto) and the STEP loop (v_cvt_f64_i32: the
    zig-zag step) of the big-launch kernel: no exec manipulation, k in an SGPR (no v_readfirstlane
    round trip), at most a handful of register moves, and instruction counts within the budget the
    per-node PMC figures of DESIGN.md section 3 correspond to (static counts of ALL blocks of a loop, the
    rare tie-rounding and global-stack blocks included); the kernel keeps 8 waves per SIMD (64 VGPRs)."""
    _, asm = artefacts
    body = _kernel_body(asm, WALK)
    found = {}
    for seg in _innermost_loops(body):
        key = "child" if any(s.startswith("v_rndne_f64") for s in seg) else \
              "step" if any(s.startswith("v_cvt_f64_i32") for s in seg) else None
        if key is None:
            continue
        valu = sum(s.startswith("v_") for s in seg)
        salu = sum(s.startswith("s_") and not s.startswith(("s_nop", "s_waitcnt", "s_cbranch", "s_branch"))
                   for s in seg)
        found[key] = dict(valu=valu, salu=salu, mov=sum(s.startswith("v_mov") for s in seg),
                          # writes of the exec mask (reading it to form vcc / scc is the uniform-branch idiom)
                          execs=sum(bool(re.match(r"s_\w+\s+exec\b", s)) or "saveexec" in s for s in seg),
                          rfl=sum(s.startswith("v_readfirstlane") for s in seg),
                          # v_readlane_b32 occupies the vector ALU for two issue slots
                          # (tests/perf/micro/valu_rates.hip): broadcasts that feed vector arithmetic go
                          # through ds_bpermute_b32 instead (the one in the CHILD chain reloads a spilled
                          # scalar in the global-stack block)
                          rl=sum(s.startswith("v_readlane") for s in seg))
    assert set(found) == {"child", "step"}, found
    for key, lim in (("child", dict(valu=44, salu=18, mov=8, rl=1)), ("step", dict(valu=22, salu=28, mov=1, rl=3))):
        f = found[key]
        assert f["execs"] == 0 and f["rfl"] == 0, (key, f)
        assert all(f[q] <= lim[q] for q in lim), (key, f)
    m = re.search(re.escape(WALK) + r"[^\n]*\n(?:.*\n)*?\s*\.vgpr_count:\s+(\d+)", asm[asm.index(".amdgpu_metadata"):])
    assert m and int(m.group(1)) <= 64, m and m.group(1)


def test_walk_kernels_use_no_scratch_memory(artefacts):
    """Every instantiation of enum_phase_kernel keeps its 64-VGPR pin WITHOUT scratch memory (round 5; 44-56 bytes per
    lane before): the per-task and per-event addresses (task columns, emitted tasks, ring records) are built from an
    opaque lane number where they are used (here_lane), so that they are not hoisted to the top of the kernel and
    kept — spilled — across the walk loops."""
    _, asm = artefacts
    meta = asm[asm.index(".amdgpu_metadata"):]
    seen = 0
    for m in re.finditer(r"\.name:\s+(_ZN5fphip17enum_phase_kernel\S+)", meta):
        blk = meta[max(0, meta.rfind("- .agpr_count", 0, m.start())):meta.find("- .agpr_count", m.end())]
        seg = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
        assert seg, m.group(1)
        assert int(seg.group(1)) == 0, (m.group(1), seg.group(1))
        seen += 1
    assert seen == 6
    assert "scratch_" not in _kernel_body(asm, WALK)
