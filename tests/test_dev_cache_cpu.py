"""fplll_amd/csrc/dev_cache.h — the library's cache of freed device blocks (hipMalloc behind a free list since the
end of round 6: the runtime's stream-ordered allocator hands out memory that is not stable under its first kernel,
tests/perf/micro/fresh_alloc_probe.hip / DESIGN.md section 6) — compiled for the HOST with counting stand-ins for
hipMalloc / hipFree / hipGetDevice / hipStreamSynchronize (tests/native/dev_cache_host.cpp): re-use of a freed block
by a request it fits (and only such), best fit, rounding, per-device lists, the wait for the owner's stream, trimming
above the cap, the retry after an out-of-memory, foreign pointers."""
import os
import subprocess

import conftest as C


def test_cache_policy_on_the_host(tmp_path):
    exe = str(tmp_path / "dev_cache_host")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(C.ROOT, "tests", "native", "dev_cache_host.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, FPHIP_DEV_CACHE_GB="1"))
    lines = r.stdout.strip().split("\n")
    assert r.returncode == 0 and len(lines) >= 20, r.stdout + r.stderr
    assert all(l.startswith("ok ") for l in lines), [l for l in lines if not l.startswith("ok ")]


def test_the_library_no_longer_calls_the_stream_ordered_allocator():
    """(a guard against its return: every device allocation of csrc/ goes through dev_mem.h)"""
    src = os.path.join(C.ROOT, "fplll_amd", "csrc")
    hits = []
    for root, _, files in os.walk(src):
        for f in files:
            if f.endswith((".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f), errors="replace").read()
                for i, line in enumerate(text.split("\n"), 1):
                    code = line.split("//")[0]
                    if "hipMallocAsync(" in code or "hipFreeAsync(" in code or "hipMallocFromPoolAsync(" in code:
                        hits.append("%s:%d" % (f, i))
    assert not hits, hits
