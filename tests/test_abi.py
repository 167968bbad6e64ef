"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/*.h declares; no compute call is made (there is no GPU here)."""
import ctypes
import glob
import os
import re

import pytest

import conftest as C


def declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(C.ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(fphip_[A-Za-z0-9_]+)\s*\(", src):
            name = m.group(1)
            # typedef'd callback types are not exported symbols
            if re.search(r"\(\s*\*\s*%s\s*\)" % name, src):
                continue
            syms.add(name)
    return sorted(syms)


def test_library_exports_every_declared_symbol():
    import fplll_amd
    lib = fplll_amd.load()
    missing = []
    for s in declared_symbols():
        try:
            getattr(lib, s)
        except AttributeError:
            missing.append(s)
    assert not missing, missing
    assert len(declared_symbols()) >= 6
    assert lib.fphip_abi_version() >= 1


def test_library_exports_nothing_undeclared():
    """The C symbols the library exports are exactly those of include/fplll_hip.h (the boundary)
    plus include/fplll_hip_debug.h (host halves of device protocols, for the CPU suite)."""
    import subprocess
    so = os.path.join(C.ROOT, "fplll_amd", "lib", "libfplll_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[2] for l in out.splitlines()
                      if len(l.split()) == 3 and l.split()[1] == "T" and l.split()[2].startswith("fphip_"))
    assert exported == declared_symbols(), sorted(set(exported) ^ set(declared_symbols()))


def test_shim_exports_plugin_entry():
    so = os.path.join(C.ROOT, "fplll_amd", "lib", "libfplll_hip_extenum.so")
    assert os.path.exists(so), "run __graft_entry__.build()"
    lib = ctypes.CDLL(so)
    lib.fplll_hip_extenum_entry.restype = ctypes.c_void_p
    assert lib.fplll_hip_extenum_entry()


def test_product_path_fails_loudly_without_gpu():
    import fplll_amd
    lib = fplll_amd.load()
    if lib.fphip_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(fplll_amd.HipError):
        fplll_amd.Context(0)


def test_product_code_never_touches_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    bad = []
    for root, _, files in os.walk(os.path.join(C.ROOT, "fplll_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                p = os.path.join(root, fn)
                txt = open(p).read()
                if re.search(r"liboracle|oracle\.h|oracle_enumerate|oracle_gso|oracle/_ref", txt):
                    if fn == "build.py":
                        continue  # build_oracle() compiles the checker; it never calls it
                    bad.append(p)
    assert not bad, bad


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/fplll_hip.h must compile as C (gcc) and as C++ (g++) with
    nothing but the standard headers, and every entry point must be declared with C linkage."""
    import subprocess
    hdr = os.path.join(C.ROOT, "include", "fplll_hip.h")
    csrc = tmp_path / "t.c"
    csrc.write_text('#include "%s"\nint main(void) { return FPHIP_OK + (int)sizeof(fphip_enum_opts) * 0; }\n' % hdr)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", str(csrc)], check=True)
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include "%s"\nint main() { return FPHIP_OK; }\n' % hdr)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", str(cpp)], check=True)
    assert 'extern "C"' in open(hdr).read()


def test_timing_accessors_are_properties():
    """`last_kernel_ms` is read as an attribute everywhere (bench.py, smoke, the tests): a decorator that
    slips onto a neighbouring method turns every such read into a bound method (round 3's regression)."""
    import ast
    for mod, cls in (("gso.py", "MatGSOBatch"), ("householder.py", "MatHouseholderBatch")):
        tree = ast.parse(open(os.path.join(C.ROOT, "fplll_amd", mod)).read())
        c = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0]
        props = {f.name for f in c.body if isinstance(f, ast.FunctionDef)
                 and any(isinstance(d, ast.Name) and d.id == "property" for d in f.decorator_list)}
        assert "last_kernel_ms" in props, (mod, props)
        assert not any(p.startswith("_") or p.startswith("get_") for p in props), (mod, props)
