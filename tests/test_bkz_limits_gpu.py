"""BKZ_MAX_TIME and BKZ_DUMP_GSO of the device BKZ drivers (fphip_gso_bkz, fphip_gso_bkz_strategies; bkz.cpp:563,
588-592 and :373-377, 456-460, 508-512, 536-539, 667-670 with dump_gso :729-790) against the real reference: the
`bkzx_*` fixtures are `ref_driver bkzfix` runs with the flag set — the reference's own dump file is in the fixture
("gso_dump"), and the device's file must hold the same entries (step, loop, the norms with their 8 digits; the time
field is the run's own) beside the same basis, status and node count."""
import glob
import json
import os

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu

DUMPS = sorted(glob.glob(os.path.join(C.GOLDEN, "bkzx_*_dump.json")))


def _nodes(i):
    return (int(i[1]) & 0xffffffff) | ((int(i[2]) & 0xffffffff) << 32)


def _run(ctx, f, batch, **kw):
    from fplll_amd.gso import MatGSOBatch
    g = MatGSOBatch(ctx, batch, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * batch))
    fl = f.get("flags", 0)
    S = f.get("strategies")
    if S is None and not (fl & 0x300):
        st, info = g.bkz(f["block_size"], f["delta"], f["eta"], f["max_loops"], bool(fl & 0x20), **kw)
    else:
        rnd, _ = C.gmp_streams_native(batch, f["rng_seed"]) if S is not None else (None, None)
        st, info = g.bkz_strategies(f["block_size"], S, rnd, f["delta"], f["eta"], max_loops=f["max_loops"],
                                    gh_bnd=bool(fl & 0x80), bounded_lll=bool(fl & 0x10), gh_factor=f["gh_factor"],
                                    auto_abort=bool(fl & 0x20), sd=bool(fl & 0x100), slide=bool(fl & 0x200), **kw)
    out = g.get_basis(0, batch)
    g.close()
    return st, info, out


@pytest.mark.parametrize("path", DUMPS, ids=lambda p: os.path.basename(p)[:-5])
def test_dump_gso_equals_the_references_file(ctx, path, tmp_path):
    f = C.load_bkz_fixture(path)
    with open(path) as fh:
        ref = json.load(fh)["gso_dump"]
    name = str(tmp_path / "gso.json")
    batch = 2
    st, info, out = _run(ctx, f, batch, dump_gso=name)
    for L in range(batch):
        assert st[L] == f["status"] and np.array_equal(out[L], f["b_out"]) and _nodes(info[L]) == f["nodes"]
        with open(name if L == 0 else name + ".%d" % L) as fh:
            text = fh.read()
        mine = json.loads(text)  # (the reference's hand-written format is valid JSON, and so is ours)
        assert [(e["step"], e["loop"]) for e in mine] == [(e["step"], e["loop"]) for e in ref]
        for a, b in zip(mine, ref):
            assert a["norms"] == b["norms"], (a["step"], a["loop"])
            assert a["time"] >= 0.0 and (a["step"] != "Input" or a["time"] == 0)
        # ... character for character apart from the time fields
        # ... in the reference's layout (bkz.cpp:745-789)
        strip = lambda s: [ln for ln in s.splitlines() if '"time"' not in ln]
        assert text.splitlines()[0] == "[" and text.splitlines()[-1] == "]"
        assert strip(text)[1:4] == [" " * 8 + "{", " " * 16 + '"step": "Input",', " " * 16 + '"loop": -1,']


def test_dump_fixtures_cover_the_three_drivers():
    steps = set()
    for p in DUMPS:
        with open(p) as fh:
            steps |= {e["step"] for e in json.load(fh)["gso_dump"]}
    assert {"Input", "Output", "End of BKZ loop", "End of SD-BKZ loop", "End of SLD loop"} <= steps


def test_max_time_zero_stops_in_front_of_the_first_tour(ctx):
    """BKZ_MAX_TIME with max_time = 0 (the reference: RED_BKZ_TIME_LIMIT, basis untouched, no enumeration)."""
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "bkzx_q40_b10_time0.json"))
    assert f["status"] == 7 and np.array_equal(f["b_in"], f["b_out"]) and f["nodes"] == 0
    st, info, out = _run(ctx, f, 3, max_time=0.0)
    assert list(st) == [7] * 3
    for L in range(3):
        assert np.array_equal(out[L], f["b_out"]) and _nodes(info[L]) == 0


def test_a_generous_max_time_changes_nothing(ctx):
    """... and with a limit the run never reaches, basis, status and node count are those of the run without the flag
    (the tours are launched one by one then, like under BKZ_AUTO_ABORT)."""
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "bkzx_q40_b10_dump.json"))
    st, info, out = _run(ctx, f, 2, max_time=3600.0)
    for L in range(2):
        assert st[L] == f["status"] and np.array_equal(out[L], f["b_out"]) and _nodes(info[L]) == f["nodes"]
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "bkzx_q64_b40_pre_gh_dump.json"))
    st, info, out = _run(ctx, f, 2, max_time=3600.0)
    for L in range(2):
        assert st[L] == f["status"] and np.array_equal(out[L], f["b_out"]) and _nodes(info[L]) == f["nodes"]
