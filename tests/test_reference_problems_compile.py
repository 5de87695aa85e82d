"""How many of the reference's problem files compile UNCHANGED against the host mirror (hipcc -fsyntax-only -x hip -I quokka_amd/host).  The sources are read in place from /root/reference (nothing is copied); skipped where the
reference tree does not exist (the GPU box).  The three problems of BASELINE.json's configs must compile; the total is reported and must
not fall below the count this round reached."""
import concurrent.futures as cf
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "quokka_amd", "host")
REF = "/root/reference/src/problems"
MUST = {"HydroBlast3D", "HydroShocktube", "RadhydroShell"}
MIN_COUNT = 61


def spacedim(pdir):
    """AMREX_SPACEDIM the reference builds this problem for (its CMakeLists.txt guards: `if (AMReX_SPACEDIM EQUAL n)`)"""
    try:
        txt = open(os.path.join(pdir, "CMakeLists.txt")).read()
    except OSError:
        return 1
    m = re.search(r"AMReX_SPACEDIM\s+(?:EQUAL|GREATER_EQUAL)\s+(\d)", txt)
    return int(m.group(1)) if m else 1


def compiles(pdir):
    srcs = sorted(glob.glob(os.path.join(pdir, "*.cpp")))
    if not srcs:
        return None
    cmd = ["/opt/rocm/bin/hipcc", "-fsyntax-only", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-I" + HOST,
           "-I" + os.path.join(HOST, ".shims"), "-I" + os.path.join(ROOT, "include"), "-I" + pdir, f"-DAMREX_SPACEDIM={spacedim(pdir)}", "-w"] + srcs
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    first = next((l for l in p.stderr.splitlines() if "error" in l), "")
    return p.returncode == 0, first


@pytest.mark.skipif(not os.path.isdir(REF), reason="no reference tree here")
def test_reference_problem_files_compile_unchanged():
    subprocess.check_call(["make", "-s", "-C", HOST, ".shims/.stamp"])
    dirs = sorted(d for d in glob.glob(os.path.join(REF, "*")) if os.path.isdir(d))
    with cf.ThreadPoolExecutor(max_workers=max(2, (os.cpu_count() or 4))) as ex:
        results = dict(zip(dirs, ex.map(compiles, dirs)))
    ok = sorted(os.path.basename(d) for d, r in results.items() if r and r[0])
    bad = {os.path.basename(d): r[1] for d, r in results.items() if r and not r[0]}
    report = os.path.join(ROOT, "gpurun_out", "reference_problems_compile.txt")
    os.makedirs(os.path.dirname(report), exist_ok=True)
    with open(report, "w") as f:
        f.write(f"{len(ok)} of {len(ok) + len(bad)} reference problem directories compile unchanged\n\nOK:\n" + "\n".join(ok) + "\n\nFAIL (first error):\n")
        f.write("\n".join(f"{k}: {v[-200:]}" for k, v in sorted(bad.items())) + "\n")
    print(f"{len(ok)} of {len(ok) + len(bad)} compile unchanged: {ok}")
    assert MUST <= set(ok), {k: bad[k] for k in MUST if k in bad}
    assert len(ok) >= MIN_COUNT


def test_unmodified_ode_integration_problem_passes_on_the_host():
    """src/problems/ODEIntegration/test_ode.cpp, unchanged, against the host mirror's adaptive RK integrator (host/compat/ode_integrate.hpp) and
    gamma-law quokka::EOS: a cooling gas integrated over ten cooling times must end within 1e-4 of T = 160.526 K.  Host code only — no GPU —, so
    this runs wherever the binary was built (bin/ref_ODEIntegration; __graft_entry__.build() builds it where the reference tree exists)."""
    exe = os.path.join(HOST, "bin", "ref_ODEIntegration")
    if not os.path.exists(exe):
        pytest.skip("bin/ref_ODEIntegration not built (needs the reference tree at build time)")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-500:]
    m = re.search(r"Relative error: ([0-9.eE+-]+)", p.stdout)
    assert m and float(m.group(1)) < 1.0e-4, p.stdout[-500:]
