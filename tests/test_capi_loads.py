"""CPU-side checks of the drop-in boundary: the C-ABI library is built, loads, and exports every symbol that
include/quokka_amd.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

from quokka_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(capi.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "quokka_amd.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char \*)\s*(qk_[A-Za-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in include/quokka_amd.h but not exported: {missing}"
    assert set(capi.DECLARED_SYMBOLS) <= declared


def test_array4_descriptor_is_amrex_compatible():
    # amrex::Array4<double>: p, jstride, kstride, nstride (Long), begin, end (Dim3), ncomp -> 64 bytes
    assert ctypes.sizeof(capi.Array4) == 64
    assert capi.Array4.p.offset == 0 and capi.Array4.jstride.offset == 8 and capi.Array4.begin.offset == 32
    assert capi.Array4.end.offset == 44 and capi.Array4.ncomp.offset == 56


def test_product_path_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under quokka_amd/ or include/ may reference it."""
    bad = []
    for base in ("quokka_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hpp", ".h", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"(import|from|include)\s+[\"<]?\.*oracle", txt) or "liboracle" in txt:
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
