#!/usr/bin/env python
"""bench.py — Mcell-updates/s of the 3-D Sedov blast on a uniform grid (BASELINE.json `metric`).

A "step" is one full RK2 hydro step of every cell (both ghost fills, both fused flux/update stages, the CFL
reduction and the dt control), counted exactly like the reference's figure of merit
(reference src/simulation.hpp:1285, 972-977).

  N = 1   BASELINE config 2: tests/blast_unigrid_256.in — 256^3 cells in 128^3 boxes, gamma = 1.4, PPM + HLLC, CFL 0.3,
          reflecting octant.  The same line carries two secondary figures measured after the timed region:
          `ncell512` (512^3 on the one GPU, the size north_star quotes the roofline target on; `--ncell 512` makes it the headline)
          and `long_run` (>= 100 steps after >= 100 warm-up steps of the 256^3 run).
  N > 1   `python bench.py --gpus N` starts its own N ranks (torch.distributed.run, one per GPU, backend nccl = RCCL); under a
          launcher that already set WORLD_SIZE it just joins.  Every rank holds 512^3 cells (64 boxes of 128^3): N = 8 is
          BASELINE config 3, the 1024^3 blast; N = 2 / 4 are 1024x512x512 / 1024x1024x512.  Boxes are cut into bricks (2x2x2
          for 8 ranks), ghost strips travel as RCCL point-to-point messages overlapped with the update of the boxes that need
          nothing remote.  `weak_256_per_gpu` is the same measurement with 256^3 cells per rank.

Output: ONE JSON line on rank 0 (see the contract in the task statement), including
  roofline     — algorithmic bytes of the dominant fused sweep kernel / its HIP-event duration vs 8 TB/s HBM, every sweep listed
  cpu_baseline — the CPU oracle (a port of the reference algorithm, NOT the reference binary) on the host cores (N = 1 only)
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import re
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# algorithmic bytes per cell-sweep of the fused PPM+HLLC sweep kernels and of the flattening pre-pass
# k_sweep_xy (round 6): the X sweep folded into the Y march — ONE kernel does the work of §8(d)'s x row (128 B) and y row (184 B); by this design's own
# count it has to move 136 B per cell (state 48 + chi, D_x, D_y, D_z 32 + accumulator out 56), reported beside the §8(d) figure
ALG_BYTES = {"k_sweep_x": 128.0, "k_sweep_y": 184.0, "k_sweep_z": 184.0, "k_sweep_xy": 312.0}
ALG_BYTES_THIS_DESIGN = {"k_sweep_xy": 136.0, "k_sweep_z": 232.0, "k_sweep_y": 184.0, "k_sweep_x": 128.0}


def dominant_kernel(carry: bool) -> str:
    """the longest kernel of a stage, timed by HIP events inside the measured region (checked against the full table of the separate pass): the fused
    X + Y sweep where it runs (carried form, no passive scalars, boxes a multiple of 64 cells wide, QK_FUSEX != 0), else the Z sweep + epilogue"""
    return "k_sweep_xy" if (carry and os.environ.get("QK_FUSEX", "1") != "0") else "k_sweep_z"
ALG_BYTES_PRE = 72.0
ALG_BYTES_STEP = 1496.0
FP64_VALU_PEAK = 256 * 64 * 2.4e9  # FP64 vector lane-instructions per second without FMA contraction (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz)
# radiation transport update kernels: doubles per cell they must move (state in / out + the face fluxes of three directions)
RAD_STREAM_WORDS = {"rad_PredictStep": 22, "rad_AddFluxesRK2": 24,
                    # qk_rad_stage_fused: X reads the state and writes the accumulator; Y reads both and writes the accumulator; Z reads state,
                    # accumulator and (stage 2) the old state and writes the state
                    "rad_sweep_x": 8, "rad_sweep_y": 12, "rad_sweep_z": 14}
# Newton-Raphson exchange kernel (radSourceCell, single group, constant opacity): VALU instructions per cell and call.  Measured: SQ_INSTS_VALU
# = 1829 per cell at the shell's 2.65 Newton iterations per solve (profiles/round3/v2_shell256_pmc_SQ.txt); the split into a part outside and
# a part inside the Newton loop follows the instruction count of the gfx950 ISA (profiles/tools/isa_count.py) scaled to that total
RAD_SOURCE_VALU_FIXED, RAD_SOURCE_VALU_PER_ITERATION = 818.0, 382.0
# qk_rad_stage_fused sweeps: measured SQ_INSTS_VALU per cell (same file), PLM
RAD_SWEEP_VALU = {"rad_sweep_x": 618.0, "rad_sweep_y": 544.0, "rad_sweep_z": 620.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ncell", type=int, default=None,
                    help="cells per dimension PER GPU: default 256 at N = 1 (blast_unigrid_256.in), 512 at N > 1 (N = 8: the 1024^3 blast); "
                         "--ncell 512 at N = 1 is the single-GPU size of the roofline target")
    ap.add_argument("--max-grid-size", type=int, default=128)
    ap.add_argument("--workload", choices=["sedov", "shell", "amr", "shell_amr"], default="sedov",
                    help="sedov = BASELINE metric (default); shell = RadhydroShell 256^3 radiation-hydro (BASELINE config 4), amr = Sedov with "
                         "max_level 2 (BASELINE config 5 geometry) and shell_amr = RadhydroShell with max_level 2 (tests/radhydro_shell_amr.in, the "
                         "reference paper's strong-scaling problem), each reported as a secondary line")
    ap.add_argument("--pow-mode", type=int, default=0, help="shell workload: 0 = libm pow(T,4) as the reference's std::pow (default), 1 = repeated multiplication")
    ap.add_argument("--rk2-mode", choices=["exact", "carry"], default="carry",
                    help="exact: flux_rk2 = 0.5 F1 + 0.5 F2 face by face as the reference (bit-identical to the oracle); carry: the RK2 average on the "
                         "cell's right-hand side (qk_hydro_stage_args::rk2_carry_rhs, <= 1e-12 relative L1; fewer bytes per step)")
    ap.add_argument("--selftest", action="store_true",
                    help="N > 1: a small blast (64^3 cells per rank in 32^3 boxes, 4 steps, early / late overlap forced on) on all ranks over RCCL, "
                         "per-box digests compared with a one-rank run of the same global problem on rank 0; prints one JSON verdict and exits "
                         "(a watchdog turns a hang into a FAIL line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (ncell512, long_run, weak_256_per_gpu)")
    ap.add_argument("--cooling-only", action="store_true", help="only the cooling128 block (its own JSON line; profiling)")
    ap.add_argument("--developed-only", action="store_true", help="only the `developed` block (its own JSON line; profiling the kernels in developed flow)")
    ap.add_argument("--long-steps", type=int, default=100, help="steps (and warm-up steps) of the long_run secondary figure")
    ap.add_argument("--cpu-ncell", type=int, default=128)
    ap.add_argument("--cpu-steps", type=int, default=400)  # at most; the sample ends after ~12 s of wall time (cpu_baseline)
    return ap.parse_args()


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py <same arguments>`"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the only mode the host driver supports (RCCL over xGMI)
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    os.execv(sys.executable, cmd)


def weak_scaled_cells(n: int, ngpus: int):
    cells = [n, n, n]
    d, r = 0, ngpus
    while r > 1:
        assert r % 2 == 0, "GPU count must be a power of two"
        cells[d % 3] *= 2
        d += 1
        r //= 2
    return cells


def usable_cores():
    """(threads to use, description): the affinity mask, capped by the cgroup CPU quota (the GPU boxes show 256 CPUs but
    grant 16 CPUs' worth of time; one thread per granted CPU is fastest — profiles/round5/cpu_leg_v1.txt: 49 / 32 / 18 M cell-updates/s with
    16 / 32 / 64 threads, the extra threads spend the quota early in each period and the whole job is throttled)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{n} schedulable CPUs"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            q = float(quota) / float(period)
            note += f", cgroup quota {q:g} CPUs"
            n = max(1, min(n, int(round(q))))
    except (OSError, ValueError):
        pass
    return n, note


def cpu_baseline(ncell: int, steps: int, seconds: float = 12.0):
    """Oracle (port of the reference algorithm, built with g++ -O3 -ffp-contract=off -fopenmp) on the host cores, in its fused form: per box one
    pass forms the primitives and flattening coefficients into L2-sized scratch, sweeps the three directions row by row with x innermost (PPM
    edges in row buffers, HLLC with selects, AVX-512 / AVX2 clones picked at load time), adds 0.5 F to the RK2 flux sum while F is in registers and
    applies the update, the limits and the dual-energy sync (oracle/hydro_fused.hpp).  Bit-identical to the operator-at-a-time form every GPU
    parity test is held to (tests/test_oracle_fused_cpu.py) — what the reference's "MPI + vectorised CPU" build does with AMReX tiling."""
    from oracle.pyoracle import SEDOV, Oracle
    threads, note = usable_cores()
    if "OMP_NUM_THREADS" in os.environ:
        threads = int(os.environ["OMP_NUM_THREADS"])
    os.environ["OMP_NUM_THREADS"] = str(threads)
    o = Oracle("direct")
    s = o.sim(SEDOV, 3, [ncell] * 3, [0, 0, 0], [1.2] * 3, [0, 0, 0], max_grid_size=[32] * 3)
    s.set_fused_fluxes(True)
    for _ in range(2):  # warm-up: page-in, and the step's temporaries enter the sim's pool
        assert s.step()
    t0 = time.perf_counter()
    done = 0
    while done < steps and time.perf_counter() - t0 < seconds:  # a bounded sample: `steps` steps or `seconds` of wall time, whichever first
        assert s.step()
        done += 1
    el = time.perf_counter() - t0
    steps = done
    value = ncell ** 3 * steps / el / 1e6
    m = re.search(r"cgroup quota ([\d.]+) CPUs", note)
    cpus = min(float(threads), float(m.group(1))) if m else float(threads)  # the CPU time actually granted: the quota when it is below the thread count
    return {"value": value, "unit": "Mcell-updates/s", "cores": threads, "kind": "port", "cpus_granted": cpus, "value_per_granted_cpu": value / cpus,
            "form": "fused per box, x-vectorised (oracle/hydro_fused.hpp); bit-identical to the operator-at-a-time oracle",
            "sample": f"Sedov {ncell}^3 in 32^3 boxes, {steps} RK2 steps in {el:.1f} s, OpenMP over boxes, {threads} threads ({note}); "
                      "CPU restatement of the reference algorithm, not the reference binary"}


# ---------------------------------------------------------------------------------------------------------------- PMC traffic
def kernel_key(name: str):
    """bench.py's name of a fused-stage kernel from its rocprofv3 display name"""
    if "k_sweep_x" in name:
        return "k_sweep_x"
    m = re.search(r"k_sweep_march<(\d)([^>]*)>", name)
    if m:
        if m.group(1) == "1" and m.group(2).replace(" ", "").endswith(",true") and m.group(2).count(",") >= 8:
            return "k_sweep_xy"  # (the ninth template argument, FUSEX)
        return {"1": "k_sweep_y", "2": "k_sweep_z"}.get(m.group(1))
    if "k_pre" in name:
        return "k_pre"
    return None


def pmc_traffic(ncell: int):
    """HBM bytes per launch of every fused-stage kernel from the NEWEST committed rocprofv3 PMC summaries of this box size
    (profiles/round*/<tag>_sedov<ncell>_pmc_{FETCH,WRITE}_SIZE.txt, written by profiles/tools/profile_bench.sh): the average over the
    kernel's dispatches of FETCH_SIZE x 2 (gfx950: wide coalesced reads are tallied at half their bytes, MI355X_MICROARCH.md §HBM) +
    WRITE_SIZE, KiB -> bytes.  bench.py cannot collect PMC counters itself; None when no summary of that size exists."""
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "round*", f"*_sedov{ncell}_pmc_FETCH_SIZE.txt")):
        w = f.replace("FETCH_SIZE", "WRITE_SIZE")
        if not os.path.exists(w):
            continue
        m = re.search(r"round(\d+)[/\\]v?(\d+)", f)
        key = (int(m.group(1)), int(m.group(2))) if m else (0, 0)
        if best is None or key > best[0]:
            best = (key, f, w)
    if best is None:
        return None, None

    def read(path, counter):
        acc = {}
        for line in open(path):
            m = re.match(rf"(.*?)\s+{counter}\s+dispatches=\s*(\d+)\s+avg=\s*([\d.]+)", line)
            if m and kernel_key(m.group(1)):
                n, v = int(m.group(2)), float(m.group(3))
                a = acc.setdefault(kernel_key(m.group(1)), [0, 0.0])
                a[0] += n
                a[1] += n * v
        return {k: v[1] / v[0] for k, v in acc.items() if v[0] > 0}

    fe, wr = read(best[1], "FETCH_SIZE"), read(best[2], "WRITE_SIZE")
    out = {k: (2.0 * fe[k] + wr[k]) * 1024.0 for k in fe if k in wr}
    return out, f"{os.path.relpath(best[1], ROOT)} (x2) + {os.path.relpath(best[2], ROOT)}"


# ---------------------------------------------------------------------------------------------------------------- BASELINE configs 4 and 5
def read_profile(ctx):
    L = ctx.L
    kernels = {}
    for k in range(L.qk_profile_num_kernels(ctx.h)):
        name, cnt, ms = C.c_char_p(), C.c_long(), C.c_double()
        L.qk_profile_get(ctx.h, k, C.byref(name), C.byref(cnt), C.byref(ms))
        kernels[name.value.decode()] = (cnt.value, ms.value)
    return kernels


def run_shell(ctx, torch, ncell, mgs, steps, warmup, pow_mode, carry=False):
    """BASELINE config 4: RadhydroShell (tests/radhydro_shell_256.in; the reference problem runs 50 steps, test_radhydro_shell.cpp:431), one update =
    hydro RK2 + all radiation substeps.  Returns the JSON object of the line (headline with --workload shell, `shell256` block otherwise)."""
    import numpy as np
    from quokka_amd.radhydro import shell_problem
    tab = np.loadtxt(os.path.join(ROOT, "tests", "golden", "dust_shell_initial_conditions.txt"), skiprows=1)
    sim = shell_problem(ctx, ncell, (tab[:, 0], tab[:, 2], tab[:, 3]), max_grid_size=mgs, pow_mode=pow_mode)
    sim.maxTimesteps_ = 10 ** 9
    sim.rk2_carry_rhs = bool(carry)  # the hydro stage pair of the shell in the headline's form of the RK2 average (--rk2-mode)
    mass0 = sum(float(sim.state_new_cc_.valid(b)[0].sum().item()) for b in range(sim.lev.nboxes))
    for _ in range(warmup):
        assert sim.step()
    L = ctx.L
    L.qk_profile_reset(ctx.h)
    L.qk_profile_enable(ctx.h, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        assert sim.step(), "radhydro advance failed"
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    L.qk_profile_enable(ctx.h, 0)
    kernels = read_profile(ctx)
    mass1 = sum(float(sim.state_new_cc_.valid(b)[0].sum().item()) for b in range(sim.lev.nboxes))
    total_cells = ncell ** 3
    per = {k: v[1] / max(v[0], 1) for k, v in sorted(kernels.items())}
    launches = {k: v[0] for k, v in sorted(kernels.items())}
    # per-kernel ceilings: the radiation update kernels stream (HBM roofline, doubles per cell: state in / out + face fluxes of three directions);
    # the Newton-Raphson exchange kernel is arithmetic-bound: its ceiling is the FP64 vector rate without FMA contraction
    # (39.3 T lane-instructions/s = 256 CUs x 64 lanes x 2.4 GHz; MI355X public spec 78.6 TFLOP/s counts an FMA as two)
    roof = {}
    for k, words in RAD_STREAM_WORDS.items():
        if k in per and per[k] > 0:
            roof[k] = {"bound": "hbm", "alg_bytes_per_cell": 8.0 * words, "ms_per_launch": per[k], "launches": launches[k],
                       "frac": 8.0 * words * total_cells / (per[k] * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if k in RAD_SWEEP_VALU:  # these are closer to the FP64 issue ceiling than to the HBM one: both fractions are reported
                roof[k]["valu_instructions_per_cell"] = RAD_SWEEP_VALU[k]
                roof[k]["frac_fp64_valu"] = RAD_SWEEP_VALU[k] * total_cells / (per[k] * 1e-3) / FP64_VALU_PEAK
    if per.get("rad_AddSourceTerms", 0) > 0:
        it = sim.rad_counters["newton_iterations"] / max(sim.rad_counters["solves"], 1)
        ops = RAD_SOURCE_VALU_FIXED + RAD_SOURCE_VALU_PER_ITERATION * it
        roof["rad_AddSourceTerms"] = {"bound": "fp64-valu", "valu_instructions_per_cell_estimate": ops, "newton_iterations_per_solve": it,
                                      "ms_per_launch": per["rad_AddSourceTerms"], "launches": launches["rad_AddSourceTerms"],
                                      "frac": ops * total_cells / (per["rad_AddSourceTerms"] * 1e-3) / FP64_VALU_PEAK}
    return {"metric": "Mcell-updates/s on RadhydroShell (one update = hydro RK2 + all radiation substeps)",
            "value": total_cells * steps / elapsed / 1e6, "unit": "Mcell-updates/s", "n_gpus": 1, "steps": steps,
            "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"RadhydroShell {ncell}^3 (tests/radhydro_shell_256.in), {mgs}^3 boxes, PLM, 1 group, kappa=20",
                       "radiation_substeps_per_step": sim.radiationCellUpdates_ / max(sim.cellUpdates_, 1),
                       "newton_iterations_per_solve": sim.rad_counters["newton_iterations"] / max(sim.rad_counters["solves"], 1),
                       "solves_per_cell_and_source_call": sim.rad_counters["solves"] / max(2 * sim.radiationCellUpdates_, 1),
                       "max_newton_iterations": sim.rad_counters["max_newton_iterations"], "pow_mode": pow_mode, "sim_time": sim.tNew_,
                       "rk2_mode": "carry" if carry else "exact",
                       "rad_energy_source": "evaluated ONCE on the host (numpy; the shell's source is constant before t0) — the reference launches "
                                            "SetRadEnergySource before every source-term call: see cxx_shell256, which does",
                       "relative_mass_change": abs(mass1 - mass0) / mass0},
            "roofline": {"kernels": roof, "note": "transport sweeps against 8 TB/s HBM (frac) and against the FP64 VALU issue rate (frac_fp64_valu; PLM instruction counts); the Newton-Raphson kernel against the FP64 VALU issue rate"},
            "kernels_ms_per_launch": per, "kernels_launches": launches,
            "reference_published_a100_1gpu": 39.04}


def run_amr(ctx, torch, dist, rank, world, ncell, steps, warmup, carry=True):
    """BASELINE config 5 geometry: tests/blast_amr_maxlev2.in (256^3 base grid, max_level 2, blocking_factor 32, subcycling + reflux).
    Several GPUs: the SAME hierarchy and the deck's own 128^3 boxes (strong scaling).  Every level has a box -> rank map of its own — a
    space-filling curve over the level's boxes, a level with fewer boxes than ranks chopped until every rank owns one (AMReX's
    refine_grid_layout; amr_simulation.py: chop_grids, distribute_sfc) — so the refined work is shared by all ranks wherever the blast sits."""
    from quokka_amd.amr_simulation import sedov_amr_problem
    mgs = min(128, ncell // 2)  # (the deck: 256^3 in 128^3 boxes; the dry run of the line scales it down)
    amr = sedov_amr_problem(ctx, ncell, 2, max_grid_size=mgs, blocking_factor=32, rank=rank, nranks=world)
    if carry:  # level 0 in the headline's form of the RK2 average, flux_rk2 formed on its coarse-fine faces only
        amr.use_carried_form(True)
    amr.overlap_children = os.environ.get("QK_AMR_OVERLAP", "1") == "1"  # (one rank: the children beside the far boxes of level 0)
    E0, M0 = amr.composite_sum(4), amr.composite_sum(0)
    hp = os.environ.get("QK_AMR_MAIN_PRIORITY", "")
    if hp:  # (experiment: the whole evolve on a stream of the given priority instead of the default stream)
        least, greatest = torch.cuda.Stream.priority_range()
        st = torch.cuda.Stream(device=ctx.device, priority={"low": least, "high": greatest}[hp])
        st.wait_stream(torch.cuda.current_stream(ctx.device))
        torch.cuda.set_stream(st)
    for _ in range(warmup):
        amr.step()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    sync()
    u0, t0 = amr.cellUpdates_, time.perf_counter()
    for _ in range(steps):
        amr.step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    E1, M1 = amr.composite_sum(4), amr.composite_sum(0)
    return {"metric": "Mcell-updates/s on 3D Sedov AMR (sum over levels, subcycled)", "value": (amr.cellUpdates_ - u0) / elapsed / 1e6,
            "unit": "Mcell-updates/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"3D Sedov blast {ncell}^3 base grid, max_level 2, blocking_factor 32, max_grid_size {mgs} "
                                   "(tests/blast_amr_maxlev2.in), subcycling + reflux",
                       "clustering": getattr(amr, "clustering", "tiles"),
                       "level0_rk2_mode": "carry + flux_rk2 on coarse-fine faces only" if getattr(amr, "rk2_carry_rhs", False) else "exact",
                       "boxes_per_level": [len(L.all_boxes) for L in amr.levels],
                       "boxes_per_level_per_rank": [[sum(1 for o in L.owner if o == r) for L in amr.levels] for r in range(world)],
                       "cells_per_level_per_rank": [[sum(int((hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1)) for (lo, hi), o in zip(L.all_boxes, L.owner)
                                                         if o == r) for L in amr.levels] for r in range(world)],
                       "cells_per_level": [amr.CountCells(l) for l in range(amr.finest_level + 1)], "sim_time": amr.tNew_,
                       "coarse_steps_total": steps + warmup, "children_beside_far_boxes": dict(amr.overlap_stats),
                       "composite_energy_relative_change": abs(E1 - E0) / abs(E0), "composite_mass_relative_change": abs(M1 - M0) / abs(M0)}}


def cxx_host_block(args, ncell):
    """bin/sedov_bench (built by __graft_entry__.build()) with the deck of BASELINE config 2: W warm-up + K timed steps, its JSON line"""
    import subprocess
    host = os.path.join(ROOT, "quokka_amd", "host")
    exe = os.path.join(host, "bin", "sedov_bench")
    if not os.path.exists(exe):
        return {"error": "quokka_amd/host/bin/sedov_bench is not built (python -c 'import __graft_entry__ as g; g.build()')"}
    cmd = [exe, os.path.join(host, "decks", "blast_unigrid_256.in"), f"amr.n_cell={ncell} {ncell} {ncell}", f"bench.warmup={args.warmup}", f"bench.steps={args.steps}",
           f"hydro.rk2_carry_rhs={1 if args.rk2_mode == 'carry' else 0}", "max_timesteps=1000000"]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = next(l for l in p.stdout.splitlines() if l.startswith("{") and "cxx_host" in l)
        blk = json.loads(line)
    except Exception as e:  # noqa: BLE001 - a secondary block must not take the headline down
        return {"error": f"{type(e).__name__}: {e}"}
    blk["rk2_mode"] = args.rk2_mode
    blk["driver"] = "quokka_amd/host/drivers/sedov_bench.cpp through QuokkaSimulation<problem_t> (C++17 host mirror), deck blast_unigrid_256.in"
    return blk


def full_run_block(args):
    """The metric as the reference counts it (src/simulation.hpp:972-981, :1285): bin/ref_HydroBlast3D — the reference's own problem file,
    compiled unchanged against the C++17 host — with the deck of BASELINE config 2 (blast_unigrid_256.in: max_timesteps = 1000), cell-updates
    divided by the wall time of the WHOLE evolve as the executable prints it, both forms of the RK2 average, plus how often the first-order flux
    correction and the retry loop ran (a line this host adds after the figure of merit)."""
    import subprocess
    host = os.path.join(ROOT, "quokka_amd", "host")
    exe = os.path.join(host, "bin", "ref_HydroBlast3D")
    if not os.path.exists(exe):
        return {"error": "quokka_amd/host/bin/ref_HydroBlast3D is not built (needs the reference tree at build time)"}
    out = {"unit": "Mcell-updates/s", "deck": "quokka_amd/host/decks/blast_unigrid_256.in (the keys of the reference's tests/blast_unigrid_256.in), unchanged: 1000 steps",
           "driver": "the reference's test_hydro3d_blast.cpp, unchanged, through QuokkaSimulation<problem_t> (C++17 host mirror); the executable's own "
                     "figure of merit: all cell-updates / wall time of evolve()"}
    for mode in (args.rk2_mode, "exact" if args.rk2_mode == "carry" else "carry"):
        cmd = [exe, os.path.join(host, "decks", "blast_unigrid_256.in"), f"hydro.rk2_carry_rhs={1 if mode == 'carry' else 0}", "plotfile_interval=-1",
               "checkpoint_interval=-1"]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=host)
            m = re.search(r"Performance figure-of-merit: ([0-9.eE+-]+) .s/zone-update \[([0-9.eE+-]+) Mupdates/s\]", p.stdout)
            c = re.search(r"qk counters: steps=(\d+) fofc_stages=(\d+) retries=(\d+) elapsed_s=([0-9.eE+-]+) sim_time=([0-9.eE+-]+)", p.stdout)
            if m is None or c is None:
                out[mode] = {"error": "no figure of merit in the output", "rc": p.returncode, "tail": p.stdout[-300:]}
                continue
            out[mode] = {"value": float(m.group(2)), "steps": int(c.group(1)), "fofc_stages": int(c.group(2)), "retries": int(c.group(3)),
                         "elapsed_s": float(c.group(4)), "sim_time": float(c.group(5)), "ms_per_step": 1e3 * float(c.group(4)) / max(int(c.group(1)), 1),
                         "energy_conservation_ok": "Energy conservation is OK." in p.stdout}
        except Exception as e:  # noqa: BLE001 - a secondary block must not take the headline down
            out[mode] = {"error": f"{type(e).__name__}: {e}"}
    if "value" in out.get(args.rk2_mode, {}):
        out["value"], out["rk2_mode"] = out[args.rk2_mode]["value"], args.rk2_mode
    return out


class ClockSampler:
    """shader clock (and board power) of the GPU this process runs on, read from the amdgpu hwmon files of its PCI device every few milliseconds
    while a timed block runs — by a separate PROCESS (a shell loop: a Python thread would take the interpreter lock from the launch loop it
    observes; measured: 5.13 -> 5.45 ms per step).  A MEASUREMENT of what round 5 inferred from SQ_BUSY_CYCLES per microsecond.  None where the
    files are missing."""

    def __init__(self, torch, device_index=0):
        self.files, self.proc, self.out = None, None, None
        try:
            p = torch.cuda.get_device_properties(device_index)
            self.bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            hw = glob.glob(f"/sys/bus/pci/devices/{self.bdf}/hwmon/hwmon*")
            if hw and os.path.exists(os.path.join(hw[0], "freq1_input")):
                self.files = (os.path.join(hw[0], "freq1_input"), os.path.join(hw[0], "power1_input"))
        except Exception:  # noqa: BLE001
            self.files = None

    def __enter__(self):
        import subprocess
        import tempfile
        if self.files is not None:
            self.out = tempfile.NamedTemporaryFile(prefix="qk_clock_", suffix=".txt", delete=False)
            loop = f'while :; do read f < {self.files[0]}; read p < {self.files[1]} 2>/dev/null || p=0; echo "$f $p"; sleep 0.004; done'
            self.proc = subprocess.Popen(["bash", "-c", loop], stdout=self.out, stderr=subprocess.DEVNULL)
        return self

    def __exit__(self, *exc):
        if self.proc is not None:
            self.proc.kill()
            self.proc.wait()
            self.proc = None

    def summary(self):
        if self.out is None:
            return None
        self.out.close()
        mhz, watts = [], []
        for line in open(self.out.name):
            w = line.split()
            if len(w) == 2 and w[0].isdigit():
                mhz.append(int(w[0]) / 1e6)
                if w[1].isdigit() and int(w[1]) > 0:
                    watts.append(int(w[1]) / 1e6)
        os.unlink(self.out.name)
        if not mhz:
            return None
        m = sorted(mhz)
        out = {"sclk_mhz_mean": sum(m) / len(m), "sclk_mhz_min": m[0], "sclk_mhz_median": m[len(m) // 2], "sclk_mhz_max": m[-1], "samples": len(m),
               "source": f"/sys/bus/pci/devices/{self.bdf}/hwmon/*/freq1_input, a shell loop in its own process during the timed steps"}
        if watts:
            out["power_w_mean"] = sum(watts) / len(watts)
        return out


def developed_block(ctx, torch, ncell, mgs, steps, warmup, carry):
    """Sedov geometry started from a DEVELOPED blast (quokka_amd.simulation.developed_state: a Mach-3 shell at 0.62 of the box edge, hot
    interior, rippled — the state tests/test_bench_geometry_gpu.py pins to the oracle bit for bit): every limiter / flattening / HLLC-fan branch
    fires in a large share of the cells, which the deck's first 1000 steps (an almost uniform ambient medium) do not do."""
    from quokka_amd.simulation import developed_state, sedov_problem
    sim = sedov_problem(ctx, ncell, max_grid_size=mgs)
    sim.maxTimesteps_ = 10 ** 9
    sim.rk2_carry_rhs = bool(carry)
    for b, (lo, hi) in enumerate(sim.my_boxes):
        sim.state_new_cc_.set_fab(b, developed_state(ncell, lo, hi))
    sim._signal_of_state_new = None
    for _ in range(warmup):
        assert sim.step()
    L = ctx.L
    L.qk_profile_reset(ctx.h)
    L.qk_profile_enable(ctx.h, 1)
    torch.cuda.synchronize()
    clk = ClockSampler(torch)
    t0 = time.perf_counter()
    with clk:
        for _ in range(steps):
            assert sim.step(), "hydro advance failed"
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    L.qk_profile_enable(ctx.h, 0)
    k = read_profile(ctx)
    rho = torch.cat([sim.state_new_cc_.valid(b)[0].reshape(-1) for b in range(sim.lev.nboxes)])
    return {"value": ncell ** 3 * steps / el / 1e6, "unit": "Mcell-updates/s", "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
            "clock": clk.summary(),
            "rk2_mode": "carry" if carry else "exact", "fofc_stages": sim.counters["fofc1_stages"] + sim.counters["fofc2_stages"],
            "retries": sim.counters["retries"], "sim_time": sim.tNew_, "density_min_max": [float(rho.min().item()), float(rho.max().item())],
            "kernels_ms_per_launch": {n: v[1] / max(v[0], 1) for n, v in sorted(k.items())},
            "initial_state": "quokka_amd.simulation.developed_state (tests/test_bench_geometry_gpu.py)"}


def cooling_block(ctx, torch, ncell=128, calls=6, cpu_cells=400000):
    """The Strang-split tabulated-cooling source (qk_cooling_tabulated, reference src/cooling/TabulatedCooling.hpp:258-317) on a multiphase medium:
    n_H 1e-3 ... 1e3 cm^-3, T 10 ... 1e8 K, log-uniform, over 2000 yr — from a fraction of a substep per cell to > 1000.  A cell costs what its
    substeps cost (two right-hand sides each: two table look-ups for the energy bounds + one Algorithm-748 solve of ~8 function values + the two
    rate look-ups), so the figure is cell-SUBSTEPS per second beside cells per second; the kernel is bound by FP64 / transcendental issue and by
    the divergence of the substep counts inside a wave, not by HBM (48 B per cell).  `cpu_port`: the oracle on a bounded sample of the same cells."""
    import ctypes as C
    import numpy as np
    from quokka_amd import capi
    from quokka_amd.cooling import CloudyTables, TabulatedCooling
    from quokka_amd.simulation import Geometry, HydroSimulation
    table = os.path.join(ROOT, "tests", "golden", "isrf_1000Go_grains.h5")
    kpc, gamma = 3.0857e21, 5.0 / 3.0
    geom = Geometry(3, [ncell] * 3, [0.0] * 3, [kpc] * 3, [1, 1, 1])
    bcs = [([capi.BC_INT_DIR] * 3, [capi.BC_INT_DIR] * 3) for _ in range(6)]
    sim = HydroSimulation(ctx, geom, capi.traits(gamma, False, 3), bcs, [min(ncell, 128)] * 3)
    cool = TabulatedCooling(sim, CloudyTables(ctx, table), T_floor=10.0)
    g = torch.Generator(device=ctx.device).manual_seed(5)
    k_B, m_H = 1.380649e-16, 1.67262192369e-24 + 9.1093837015e-28
    for b in range(sim.lev.nboxes):
        v = sim.state_new_cc_.valid(b)
        shape = v[0].shape
        rho = 10 ** (torch.rand(shape, generator=g, device=ctx.device, dtype=torch.float64) * 6.0 - 27.0)
        T = 10 ** (torch.rand(shape, generator=g, device=ctx.device, dtype=torch.float64) * 7.0 + 1.0)
        E = cool.evaluate(1, rho.reshape(-1), T.reshape(-1)).reshape(shape)  # ComputeEgasFromTgas
        v[0], v[4], v[5] = rho, E, E
        v[1:4] = 0.0
    sim.state_old_cc_.copy_from(sim.state_new_cc_)
    dt = 3.15e7 * 2.0e3
    L = ctx.L
    assert cool(sim.state_new_cc_, 0.0, dt)  # warm-up
    navg, nmax = cool.last
    L.qk_profile_reset(ctx.h)
    L.qk_profile_only(ctx.h, None)
    L.qk_profile_enable(ctx.h, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        sim.state_new_cc_.copy_from(sim.state_old_cc_)  # the same work every call
        assert cool(sim.state_new_cc_, 0.0, dt)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    L.qk_profile_enable(ctx.h, 0)
    k = read_profile(ctx)
    cnt, ms = k.get("cooling_tabulated", (0, 0.0))
    kernel_ms = ms / max(cnt, 1)
    cells = ncell ** 3
    out = {"value": cells / (kernel_ms * 1e-3) / 1e6 if kernel_ms > 0 else None, "unit": "Mcells/s (one source call)", "kernel_ms": kernel_ms, "calls": cnt,
           "wall_ms_per_call_with_state_copy_and_readback": el / calls * 1e3, "substeps_per_cell_avg": navg, "substeps_max": nmax,
           "Msubsteps_per_s": cells * navg / (kernel_ms * 1e-3) / 1e6 if kernel_ms > 0 else None, "dt_yr": 2000.0,
           "workload": f"{ncell}^3 cells, n_H 1e-3..1e3 cm^-3, T 10..1e8 K log-uniform, Cloudy table isrf_1000Go_grains.h5, T_floor 10 K", "bound": "FP64 issue / divergence"}
    try:  # the oracle on a sample of the same distribution (cells are independent: no geometry)
        from mini_hdf5 import cloudy_file_arrays  # noqa: E402  (tests/ on the path below)
    except ImportError:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from mini_hdf5 import cloudy_file_arrays
    from oracle.pyoracle import OracleCloudy
    orc = OracleCloudy(cloudy_file_arrays(table))
    r = np.random.default_rng(5)
    rho_c, T_c = 10 ** r.uniform(-27.0, -21.0, cpu_cells), 10 ** r.uniform(1.0, 8.0, cpu_cells)
    U = np.zeros((6, cpu_cells))
    U[0] = rho_c
    U[4] = U[5] = orc.evaluate(orc.EGAS_FROM_TGAS, rho_c, T_c, gamma)
    threads, note = usable_cores()
    t0 = time.perf_counter()
    _, ns = orc.compute_cooling(U, gamma, dt, 10.0)
    elc = time.perf_counter() - t0
    out["cpu_port"] = {"value": cpu_cells / elc / 1e6, "unit": "Mcells/s", "Msubsteps_per_s": float(ns.sum()) / elc / 1e6, "cores": threads,
                       "sample": f"{cpu_cells} cells of the same distribution in {elc:.1f} s, OpenMP over cells ({note})", "substeps_per_cell_avg": float(ns.mean())}
    return out


def cxx_shell_block(args, steps=22):
    """BASELINE config 4 through the C++17 host: the reference's OWN problem file (src/problems/RadhydroShell, compiled unchanged against the host
    mirror by __graft_entry__.build(), where the reference tree exists) with the deck of the config; the figure of merit the executable prints
    (AMRSimulation::evolve: all steps of the run, the first ones included).  `value`: SetRadEnergySource evaluated before every source-term call
    as the reference does (QuokkaSimulation.hpp:1866-1873); `source_evaluated_once`: the same run with this host's extension
    radiation.source_is_time_independent = 1 (the shell's source does not change before t0: same result, one launch instead of 2 per substep)."""
    import re
    import subprocess
    host = os.path.join(ROOT, "quokka_amd", "host")
    exe = os.path.join(host, "bin", "ref_RadhydroShell")
    if not os.path.exists(exe):
        return {"error": "quokka_amd/host/bin/ref_RadhydroShell is not built (needs the reference tree at build time)"}
    gold = os.path.join(ROOT, "tests", "golden", "dust_shell_initial_conditions.txt")
    import tempfile
    vals = {}
    with tempfile.TemporaryDirectory() as tmp:
        import shutil
        shutil.copy(gold, os.path.join(tmp, "initial_conditions.txt"))  # the problem opens ./initial_conditions.txt
        for flag, nsteps in ((0, steps), (1, steps), (0, 4)):  # (the last run: the first 4 steps alone, for the steady window)
            cmd = [exe, os.path.join(host, "decks", "radhydro_shell_256.in"), f"max_timesteps={nsteps}", "plotfile_interval=-1", "checkpoint_interval=-1",
                   f"radiation.source_is_time_independent={flag}", f"hydro.rk2_carry_rhs={1 if args.rk2_mode == 'carry' else 0}"]
            try:
                p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=tmp)
                m = re.search(r"Performance figure-of-merit: ([0-9.eE+-]+) .s/zone-update \[([0-9.eE+-]+) Mupdates/s\]", p.stdout)
                if m is None:
                    return {"error": "no figure of merit in the output", "tail": p.stdout[-300:]}
                vals[(flag, nsteps)] = float(m.group(2))
            except Exception as e:  # noqa: BLE001 - a secondary block must not take the headline down
                return {"error": f"{type(e).__name__}: {e}"}
    vals[0], vals[1] = vals[(0, steps)], vals[(1, steps)]
    cells = 256.0 ** 3
    t_full, t_head = cells * steps / (vals[0] * 1e6), cells * 4 / (vals[(0, 4)] * 1e6)
    steady = cells * (steps - 4) / max(t_full - t_head, 1e-9) / 1e6
    return {"value": vals[0], "unit": "Mcell-updates/s", "steps": steps, "rk2_mode": args.rk2_mode,
            "steady_value": steady, "steady_window": f"steps 5 .. {steps} (two runs): start-up — code-object loading, plans, the first steps' fewer substeps — left out, as in the Python block",
            "source_evaluated_once": {"value": vals[1], "flag": "radiation.source_is_time_independent=1 (an extension of this host, not a reference key)"},
            "driver": "the reference's test_radhydro_shell.cpp, unchanged, through QuokkaSimulation<problem_t> (C++17 host mirror), deck radhydro_shell_256.in, "
                      "SetRadEnergySource evaluated before every source-term call as in the reference; the executable's own figure of merit over all steps of the run"}


def cxx_amr_block(args, steps=55, skip=5):
    """BASELINE config 5 geometry through the C++17 host: the reference's unchanged test_hydro3d_blast.cpp on the deck of the config
    (blast_amr_maxlev2.in).  `value`: the executable's own figure of merit over the whole evolve of `steps` coarse steps (start-up — code-object
    loading at the first launches, plans, the first step's retry — included: ~50 ms of a 500 ms run); `steady_value`: the same window bench.py's
    Python block times (coarse steps skip + 1 .. steps), from the difference of two runs."""
    import subprocess
    host = os.path.join(ROOT, "quokka_amd", "host")
    exe = os.path.join(host, "bin", "ref_HydroBlast3D")
    if not os.path.exists(exe):
        return {"error": "quokka_amd/host/bin/ref_HydroBlast3D is not built (needs the reference tree at build time)"}
    carry = 1 if args.rk2_mode == "carry" else 0

    def run(n):
        cmd = [exe, os.path.join(host, "decks", "blast_amr_maxlev2.in"), f"max_timesteps={n}", f"hydro.rk2_carry_rhs={carry}", "plotfile_interval=-1", "checkpoint_interval=-1"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=host)
        m = re.search(r"Performance figure-of-merit: ([0-9.eE+-]+) .s/zone-update \[([0-9.eE+-]+) Mupdates/s\]", p.stdout)
        if m is None:
            raise RuntimeError("no figure of merit in the output: " + p.stdout[-300:])
        lv = re.findall(r"Zone-updates on level (\d): (\d+) \((\d+) grids\)", p.stdout)
        sp = re.search(r"speculative coarse steps: overlapped=(\d+) rolled_back=(\d+)", p.stdout)
        updates = sum(int(x[1]) for x in lv)
        return {"fom": float(m.group(2)), "updates": updates, "seconds": updates / (float(m.group(2)) * 1e6), "levels": [int(x[1]) for x in lv],
                "ok": "Energy conservation is OK." in p.stdout, "spec": (int(sp.group(1)), int(sp.group(2))) if sp else None}

    try:
        full = run(steps)
        head = run(skip)
    except Exception as e:  # noqa: BLE001 - a secondary block must not take the headline down
        return {"error": f"{type(e).__name__}: {e}"}
    steady = (full["updates"] - head["updates"]) / max(full["seconds"] - head["seconds"], 1e-9) / 1e6
    return {"value": full["fom"], "unit": "Mcell-updates/s", "coarse_steps": steps, "steady_value": steady, "steady_window": f"coarse steps {skip + 1} .. {steps} (two runs)",
            "level0_rk2_mode": "carry + flux_rk2 on coarse-fine faces only" if carry else "exact",
            "zone_updates_per_level": full["levels"], "energy_conservation_ok": full["ok"],
            "speculative_coarse_steps": None if full["spec"] is None else {"overlapped": full["spec"][0], "rolled_back": full["spec"][1]},
            "driver": "the reference's test_hydro3d_blast.cpp, unchanged, through QuokkaSimulation<problem_t> + AmrDriver (C++17 host mirror), deck blast_amr_maxlev2.in; "
                      "value = the executable's own figure of merit over the whole evolve"}


def compact(block, keep=("value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline", "kernels_ms_per_launch")):
    """a secondary block of the default line: the figures of a workload's own line without the contract boilerplate"""
    return {k: block[k] for k in keep if k in block}



# ---------------------------------------------------------------------------------------------------------------- N > 1 self-test
def selftest(ctx, torch, dist, rank, world, carry, inline=False):
    """the first thing to run on a multi-GPU lease: does the RCCL path (strip pack -> P2P send / recv -> unpack, early / late schedule, fused
    all-reduce of dt and counters) reproduce the one-rank state bit for bit?  A hang becomes a FAIL line after `limit` seconds.
    inline: called by the N > 1 bench run before its timed region — the verdict goes into the bench line (`selftest`) instead of on its own
    line, and the process group stays up."""
    import hashlib
    import signal
    from quokka_amd.simulation import sedov_problem
    limit = 240

    def on_alarm(signum, frame):
        print(json.dumps({"selftest": "FAIL", "reason": f"rank {rank}: no result after {limit} s (hang in the exchange?)", "n_gpus": world}), flush=True)
        os._exit(3)
    signal.signal(signal.SIGALRM, on_alarm)
    signal.alarm(limit)
    ncell, mgs, steps = 64, 32, 4
    n_cell = weak_scaled_cells(ncell, world)

    def digests(sim):
        return {tuple(lo): hashlib.sha256(v.tobytes()).hexdigest() for (lo, hi), v in zip(sim.my_boxes, sim.gather_valid_local())}

    sim = sedov_problem(ctx, ncell, max_grid_size=mgs, rank=rank, nranks=world, n_cell=n_cell)
    sim.rk2_carry_rhs = carry
    sim.min_overlap_cells = 1  # the early / late split even on this small problem
    dts = []
    for _ in range(steps):
        assert sim.step(), "hydro advance failed"
        dts.append(sim.dt_)
    mine = digests(sim)
    groups = sim.overlap_groups()
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, dts))
    verdict = None
    if rank == 0:
        one = sedov_problem(ctx, ncell, max_grid_size=mgs, rank=0, nranks=1, n_cell=n_cell)
        one.rk2_carry_rhs = carry
        dts1 = []
        for _ in range(steps):
            assert one.step()
            dts1.append(one.dt_)
        want = digests(one)
        got = {}
        for d, _ in gathered:
            got.update(d)
        bad = sorted(k for k in want if got.get(k) != want[k])
        dt_ok = all(g[1] == dts1 for g in gathered)
        verdict = {"selftest": "PASS" if (not bad and dt_ok and len(got) == len(want)) else "FAIL", "n_gpus": world, "backend": dist.get_backend(),
                   "workload": f"3D Sedov {n_cell[0]}x{n_cell[1]}x{n_cell[2]}, {mgs}^3 boxes, {steps} steps, rk2_mode {'carry' if carry else 'exact'}",
                   "boxes": len(want), "boxes_differing_from_one_rank": len(bad), "first_differing_boxes": bad[:4], "dt_equal_on_all_ranks": dt_ok,
                   "peers_rank0": len(sim.ghost.peers), "early_late_boxes_rank0": None if groups is None else [len(groups[0][1]), len(groups[1][1])]}
        if not inline:
            print(json.dumps(verdict), flush=True)
    signal.alarm(0)
    dist.barrier()
    if not inline:
        dist.destroy_process_group()
    return verdict


# ---------------------------------------------------------------------------------------------------------------- Sedov runs
def run_sedov(ctx, torch, dist, rank, world, ncell, mgs, steps, warmup, profile=True, carry=False):
    """build the problem, `warmup` untimed steps, then EXACTLY `steps` timed steps between barrier + synchronize pairs; max over ranks"""
    from quokka_amd.simulation import sedov_problem
    n_cell = weak_scaled_cells(ncell, world)
    sim = sedov_problem(ctx, ncell, max_grid_size=mgs, rank=rank, nranks=world, n_cell=n_cell)
    sim.maxTimesteps_ = 10 ** 9
    sim.rk2_carry_rhs = bool(carry)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        assert sim.step()
    L = ctx.L
    if profile:
        # HIP events on the launch stream around the DOMINANT kernel only (two event records per timed launch cost ~6 us of GPU time: with all 14
        # launches of a step timed, the events themselves were 1.5 % of the step); the other kernels are timed in a separate pass (main: repeats)
        L.qk_profile_reset(ctx.h)
        L.qk_profile_only(ctx.h, dominant_kernel(bool(carry)).encode() if profile is True else None)  # (profile="all": every kernel — the secondary blocks)
        L.qk_profile_enable(ctx.h, 1)
    if world > 1:
        sim.ghost.exposed_events = []  # (two event records per fill: how long the compute stream stalls for the peers' strips)
    barrier()
    clk = ClockSampler(torch, ctx.device.index if hasattr(ctx.device, "index") and ctx.device.index is not None else 0)
    t0 = time.perf_counter()
    with clk:
        for _ in range(steps):
            assert sim.step(), "hydro advance failed"
        barrier()
    elapsed = time.perf_counter() - t0
    sim.clock_during_timed_steps = clk.summary()
    if profile:
        L.qk_profile_enable(ctx.h, 0)
        L.qk_profile_only(ctx.h, None)
    if world > 1:
        from quokka_amd import comm
        ev = sim.ghost.exposed_events
        sim.ghost.exposed_events = None
        waits = [a.elapsed_time(b) for a, b in ev]
        exposed = sum(waits) / max(len(waits), 1)
        edges = [0.05, 0.2, 1.0, 5.0]  # ms
        hist = [sum(1 for w in waits if (lo <= w < hi)) for lo, hi in zip([0.0] + edges, edges + [float("inf")])]
        sent = float(sum(sbuf.numel() * sbuf.element_size() for _, _, sbuf, _ in sim.ghost.peers))
        t = torch.tensor([elapsed, exposed, sent], dtype=torch.float64, device=ctx.device)
        comm.all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        sim.exchange_stats = {"fills_timed": len(ev), "exposed_ms_per_fill_max_over_ranks": float(t[1].item()), "exposed_ms_per_fill_rank0": exposed,
                              "exposed_ms_histogram_rank0": {"edges_ms": edges, "fills": hist, "max_ms": max(waits) if waits else 0.0},
                              "bytes_sent_per_fill_max_over_ranks": float(t[2].item()), "bytes_sent_per_fill_rank0": sent,
                              "note": "exposed = time the compute stream waits for the peers' strips after the boxes that need nothing remote were "
                                      "advanced (HIP events around the wait); bytes = packed ghost strips one rank sends per fill, all components"}
    kernels = {}
    if profile:  # per-kernel HIP-event durations recorded on the launch stream during the timed region
        for k in range(L.qk_profile_num_kernels(ctx.h)):
            name, cnt, ms = C.c_char_p(), C.c_long(), C.c_double()
            L.qk_profile_get(ctx.h, k, C.byref(name), C.byref(cnt), C.byref(ms))
            kernels[name.value.decode()] = (cnt.value, ms.value)
    return sim, n_cell, elapsed, kernels


def roofline_of(kernels, cells_local, total_cells, steps, elapsed, world, ncell, mgs, kernels_all=None):
    """kernels: HIP-event durations of the TIMED region (the dominant kernel only, see run_sedov); kernels_all: every kernel, timed in a separate
    pass over the same number of steps (None: `kernels` holds them all)"""
    def table(ks):
        out = {}
        for k, alg in ALG_BYTES.items():
            if k in ks and ks[k][0] > 0:
                avg_s = ks[k][1] / ks[k][0] * 1e-3
                out[k] = {"alg_bytes_per_cell": alg, "alg_bytes_per_cell_this_design": ALG_BYTES_THIS_DESIGN.get(k), "avg_launch_ms": avg_s * 1e3, "launches": ks[k][0],
                          "achieved_GBs": alg * cells_local / avg_s / 1e9, "frac": alg * cells_local / avg_s / 1e9 / HBM_PEAK_GBS}
        return out
    timed = table(kernels)
    if not timed:
        return None
    dom = max(timed, key=lambda k: timed[k]["avg_launch_ms"])
    sweeps = table(kernels_all) if kernels_all is not None else dict(timed)
    if kernels_all is not None:
        longest = max(sweeps, key=lambda k: sweeps[k]["avg_launch_ms"]) if sweeps else dom
        sweeps[dom] = dict(timed[dom], separate_pass_avg_launch_ms=sweeps.get(dom, {}).get("avg_launch_ms"))
        kernels = dict(kernels_all, **{dom: kernels[dom]})
    else:
        longest = dom
    traffic, source = pmc_traffic(ncell) if mgs == 128 else (None, None)
    pre_ms = sum(v[1] / max(v[0], 1) for k, v in kernels.items() if k.startswith("k_pre"))
    d = sweeps[dom]
    r = {"bound": "hbm", "kernel": dom, "achieved": d["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac"],
         "traffic": traffic.get(dom) if traffic else None, "traffic_unit": "bytes per launch (PMC, FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": source,
         "alg_bytes_per_launch": d["alg_bytes_per_cell"] * cells_local, "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"],
         "sweeps": sweeps, "longest_kernel_of_the_full_table": longest,
         "sweeps_note": None if kernels_all is None else f"{dom}: HIP events over the timed region; the other kernels: HIP events over a separate pass of the same "
                                                         "number of steps right after it (every event pair costs ~6 us of GPU time: timing all 14 launches of a "
                                                         "step inside the measured region made the measurement 1.5 % slower than the run)",
         "pre_pass": {"alg_bytes_per_cell": ALG_BYTES_PRE, "ms_per_stage": pre_ms,
                      "frac": (ALG_BYTES_PRE * cells_local / (pre_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pre_ms > 0 else None},
         "traffic_all_kernels": traffic,
         "whole_step": {"alg_bytes_per_cell_update": ALG_BYTES_STEP,
                        "achieved_GBs": ALG_BYTES_STEP * total_cells * steps / elapsed / 1e9 / world,
                        "frac": ALG_BYTES_STEP * total_cells * steps / elapsed / 1e9 / world / HBM_PEAK_GBS},
         "all_kernels_ms_per_launch": {k: v[1] / max(v[0], 1) for k, v in sorted(kernels.items())}}
    # every kernel of the step is in the table (the dominant one from the timed region, the others from the separate pass): what the host adds
    if kernels_all is not None and steps > 0:
        per_step = sum(v[1] for v in kernels.values()) / steps
        r["step_minus_sum_of_kernels_ms"] = elapsed * 1e3 / steps - per_step
    return r


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # QK_BENCH_ONE_GPU_TEST=1 (tests/test_multirank_one_gpu.py only): all ranks share cuda:0 and talk over gloo with host-staged buffers
    # (quokka_amd/comm.py) — RCCL refuses two ranks on one device.  Never a measurement: refused unless --selftest.
    share = os.environ.get("QK_BENCH_ONE_GPU_TEST") == "1"
    if share:  # never a measurement: --selftest, or a DRY RUN of the N > 1 line on a small problem (marked `dry_run` in the line)
        local_rank = 0
        if not args.selftest and args.ncell is None:
            args.ncell, args.max_grid_size = 64, 32
    if torch.cuda.device_count() < (local_rank + 1):
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPUs are visible (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=600))
        world = dist.get_world_size()  # n_gpus below = the ranks RCCL actually connected

    from quokka_amd.multifab import Context

    ctx = Context(local_rank)
    if args.cooling_only:
        print(json.dumps({"cooling128": cooling_block(ctx, torch)}), flush=True)
        return
    if args.developed_only:
        print(json.dumps({"developed": developed_block(ctx, torch, args.ncell or 256, args.max_grid_size, args.steps, args.warmup, args.rk2_mode == "carry")}), flush=True)
        return
    if args.selftest:
        if world == 1:
            raise SystemExit("--selftest compares N > 1 ranks with one rank: use --gpus N")
        v = selftest(ctx, torch, dist, rank, world, carry=(args.rk2_mode == "carry"))
        sys.exit(0 if (rank != 0 or v["selftest"] == "PASS") else 1)
    ncell = args.ncell if args.ncell is not None else (256 if world == 1 else 512)
    selftest_verdict = None
    if world > 1 and args.workload == "sedov":
        # the exchange path proves itself before it is timed: N ranks against one rank, bit for bit (a hang becomes a FAIL line, see selftest)
        selftest_verdict = selftest(ctx, torch, dist, rank, world, carry=(args.rk2_mode == "carry"), inline=True)
    if args.workload == "shell_amr":
        import numpy as np
        from quokka_amd.amr_simulation import shell_amr_problem
        assert world == 1, "shell_amr: one GPU here (the multi-rank path is covered by tests/test_multirank_one_gpu.py)"
        # (the deck's 256^3 base grid refines the whole shell twice: > 10^8 cells on level 2, which with this driver's per-level work arrays
        # does not fit one GPU — the reference ran it on 4 to 32; the one-GPU figure is quoted on a 128^3 base grid)
        ncell = args.ncell if args.ncell is not None else 128
        tab = np.loadtxt(os.path.join(ROOT, "tests", "golden", "dust_shell_initial_conditions.txt"), skiprows=1)
        amr = shell_amr_problem(ctx, ncell, 2, (tab[:, 0], tab[:, 2], tab[:, 3]), max_grid_size=128, blocking_factor=32, pow_mode=args.pow_mode)
        for _ in range(args.warmup):
            amr.step()
        torch.cuda.synchronize()
        u0, t0 = amr.cellUpdates_, time.perf_counter()
        for _ in range(args.steps):
            amr.step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        print(json.dumps({"metric": "Mcell-updates/s on RadhydroShell AMR (sum over levels, subcycled; one update = hydro RK2 + all radiation substeps)",
                          "value": (amr.cellUpdates_ - u0) / elapsed / 1e6, "unit": "Mcell-updates/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                          "data": "synthetic",
                          "config": {"workload": f"RadhydroShell {ncell}^3 base grid, max_level 2, blocking_factor 32, max_grid_size 128 (tests/radhydro_shell_amr.in)",
                                     "boxes_per_level": [len(L.all_boxes) for L in amr.levels], "cells_per_level": [amr.CountCells(l) for l in range(amr.finest_level + 1)],
                                     "pow_mode": args.pow_mode, "sim_time": amr.tNew_},
                          "reference_published_v100_4gpu": 19.82}), flush=True)
        return
    if args.workload == "amr":
        out = run_amr(ctx, torch, dist, rank, world, args.ncell if args.ncell is not None else 256, args.steps, args.warmup)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.workload == "shell":
        assert world == 1, "the shell line is single-GPU"
        out = run_shell(ctx, torch, args.ncell if args.ncell is not None else 256, args.max_grid_size, args.steps, args.warmup, args.pow_mode,
                        carry=(args.rk2_mode == "carry"))
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_shell()
        print(json.dumps(out), flush=True)
        return

    # ------------------------------------------------------------------------------------------------------------ Sedov (headline)
    mgs = args.max_grid_size
    sim, n_cell, elapsed, kernels = run_sedov(ctx, torch, dist, rank, world, ncell, mgs, args.steps, args.warmup, carry=(args.rk2_mode == "carry"))
    cells_local = sim.lev.num_cells()
    total_cells = n_cell[0] * n_cell[1] * n_cell[2]
    groups = sim.overlap_groups() if world > 1 else None
    out = {
        "metric": "Mcell-updates/s on 3D Sedov unigrid", "value": total_cells * args.steps / elapsed / 1e6, "unit": "Mcell-updates/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"3D Sedov blast {n_cell[0]}x{n_cell[1]}x{n_cell[2]} unigrid (tests/blast_unigrid_256.in scaled to {ncell}^3 cells per GPU), "
                               f"{mgs}^3 boxes, PPM+HLLC RK2, gamma=1.4, CFL 0.3, reflecting octant",
                   "cells_per_gpu": ncell ** 3, "boxes_per_gpu": sim.lev.nboxes, "parallelism": f"box-decomposition x{world}",
                   "ghost_exchange": None if world == 1 else dict({"backend": dist.get_backend(), "peers_rank0": len(sim.ghost.peers),
                                                                   "overlap_early_late_boxes_rank0": None if groups is None else [len(groups[0][1]), len(groups[1][1])]},
                                                                  **getattr(sim, "exchange_stats", {})),
                   "rk2_mode": args.rk2_mode,
                   "prim_handoff": {"on": sim._prim_handoff_applies(), "dropped_attempts": sim.counters.get("prim_handoff_dropped", 0),
                                    "what": "stage 1's final sweep stores the primitives of the intermediate state, stage 2's pre-pass and sweeps read "
                                            "them (qk_hydro_stage_args::prim_out / prim_in): same bytes, same bits, 4.9 conversions per cell less"},
                   "fofc_stages": sim.counters["fofc1_stages"] + sim.counters["fofc2_stages"], "retries": sim.counters["retries"],
                   "sim_time": sim.tNew_,
                   "x_sweep_in_y_march": os.environ.get("QK_FUSEX", "1") != "0" and args.rk2_mode == "carry" and mgs % 64 == 0,
                   "clock": getattr(sim, "clock_during_timed_steps", None),
                   "note": "N = 1 runs BASELINE config 2 (256^3); N > 1 runs 512^3 cells per GPU (N = 8: config 3, 1024^3); "
                           "weak_256_per_gpu is the same geometry as N = 1 on every rank"},
        "roofline": None,  # (filled below, after the pass that times every kernel)
        # context only (other hardware, reference implementation): paper/performance_a100.csv:2 = 254.05 Mzones/s on 1x A100
        "reference_published_a100_1gpu": 254.05,
    }
    if world > 1:
        out["selftest"] = selftest_verdict  # (rank 0's; None on the other ranks, which do not print)
    if share:
        out["dry_run"] = f"{world} ranks share cuda:0 and talk over gloo with host-staged buffers: the SHAPE of the N > 1 line on a small problem, never a measurement"
    # the same measurement twice more on the same run (the box's HBM rate drifts by several percent within minutes: profiles/round4/README.md)
    reps = []
    kernels_all = None
    for rep in range(2):
        if rep == 0:  # this pass times EVERY kernel with HIP events (the table of the roofline object); the second one none
            ctx.L.qk_profile_reset(ctx.h)
            ctx.L.qk_profile_only(ctx.h, None)
            ctx.L.qk_profile_enable(ctx.h, 1)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            assert sim.step(), "hydro advance failed"
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        reps.append(time.perf_counter() - t0)
        if rep == 0:
            ctx.L.qk_profile_enable(ctx.h, 0)
            kernels_all = read_profile(ctx)
    out["roofline"] = roofline_of(kernels, cells_local, total_cells, args.steps, elapsed, world, ncell, mgs, kernels_all=kernels_all)
    if world > 1:
        t = torch.tensor(reps, dtype=torch.float64, device=ctx.device)
        from quokka_amd import comm
        comm.all_reduce(t, dist.ReduceOp.MAX)
        reps = t.tolist()
    out["repeats"] = {"values": [total_cells * args.steps / r / 1e6 for r in reps], "unit": "Mcell-updates/s", "steps_each": args.steps,
                      "note": "the timed region repeated twice on the continuing run, after `value` was taken: the first with HIP events around every "
                              "kernel (the roofline table), the second with none"}
    del sim
    torch.cuda.empty_cache()
    if not args.no_secondary:
        if world == 1 and ncell == 256:
            # (a) developed-run figure: the default timed region sits at sim-time ~6e-5; >= 100 + 100 steps moves the shock further out
            carry = args.rk2_mode == "carry"  # every secondary Sedov figure uses the headline's form of the RK2 average
            simL, _, elL, _ = run_sedov(ctx, torch, dist, rank, world, 256, mgs, args.long_steps, args.long_steps, profile=False, carry=carry)
            out["long_run"] = {"value": 256 ** 3 * args.long_steps / elL / 1e6, "unit": "Mcell-updates/s", "steps": args.long_steps, "warmup": args.long_steps,
                               "ms_per_step": elL / args.long_steps * 1e3, "sim_time": simL.tNew_,
                               "fofc_stages": simL.counters["fofc1_stages"] + simL.counters["fofc2_stages"]}
            del simL
            torch.cuda.empty_cache()
            # (b) 512^3 on the one GPU: the size of north_star's roofline target
            s5, n5, el5, k5 = run_sedov(ctx, torch, dist, rank, world, 512, mgs, 8, 2, carry=carry, profile="all")
            out["ncell512"] = {"rk2_mode": args.rk2_mode, "value": 512 ** 3 * 8 / el5 / 1e6, "unit": "Mcell-updates/s", "steps": 8, "warmup": 2, "ms_per_step": el5 / 8 * 1e3,
                               "boxes": s5.lev.nboxes, "roofline": roofline_of(k5, s5.lev.num_cells(), 512 ** 3, 8, el5, 1, 512, mgs)}
            del s5
            torch.cuda.empty_cache()
            # (c) the other form of the RK2 average, same run (headline: --rk2-mode, default carry; `rk2_other_mode`: the remaining one)
            other = "exact" if args.rk2_mode == "carry" else "carry"
            sO, _, elO, kO = run_sedov(ctx, torch, dist, rank, world, 256, mgs, args.steps, args.warmup, carry=(other == "carry"), profile="all")
            out["rk2_other_mode"] = {"rk2_mode": other, "value": 256 ** 3 * args.steps / elO / 1e6, "unit": "Mcell-updates/s", "ms_per_step": elO / args.steps * 1e3,
                                     "kernels_ms_per_launch": {k: v[1] / max(v[0], 1) for k, v in sorted(kO.items())},
                                     "note": "exact = flux_rk2 = 0.5 F1 + 0.5 F2 face by face (bit-identical to the CPU oracle); carry = the average taken on the "
                                             "cell's right-hand side (<= 1e-12 relative L1 per conserved component, tests/test_hydro_step_gpu.py, "
                                             "tests/test_bench_geometry_gpu.py)"}
            del sO
            torch.cuda.empty_cache()
            # (c2) the same measurement through the C++17 host mirror (quokka_amd/host: QuokkaSimulation<problem_t> as a problem file drives it;
            # builder-authored driver quokka_amd/host/drivers/sedov_bench.cpp, its own process)
            out["cxx_host"] = cxx_host_block(args, 256)
            # (c3) the metric as the reference counts it: the unchanged problem file + deck, 1000 steps, the executable's own figure of merit
            out["full_run"] = full_run_block(args)
            # (c4) developed flow: 50 steps from a strong shell at 0.62 of the box edge
            out["developed"] = developed_block(ctx, torch, 256, mgs, 50, 3, carry)
            torch.cuda.empty_cache()
            # (d) BASELINE config 4 at its full size: RadhydroShell 256^3, the 50 steps the reference problem runs (test_radhydro_shell.cpp:431)
            out["shell256"] = compact(run_shell(ctx, torch, 256, 128, 50, 2, 0, carry=(args.rk2_mode == "carry")))
            out["cxx_shell256"] = cxx_shell_block(args)
            torch.cuda.empty_cache()
            # (e) BASELINE config 5 geometry at its full size on the one GPU: blast_amr_maxlev2.in, 256^3 base grid + 2 levels
            out["amr_maxlev2"] = compact(run_amr(ctx, torch, dist, rank, world, 256, 50, 5, carry=(args.rk2_mode == "carry")))
            torch.cuda.empty_cache()
            out["cxx_amr_maxlev2"] = cxx_amr_block(args)
            # (f) the Strang-split cooling source on a multiphase medium
            try:
                out["cooling128"] = cooling_block(ctx, torch)
            except Exception as e:  # noqa: BLE001 - a secondary block must not take the headline down
                out["cooling128"] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
        elif world > 1:
            if ncell not in (256, 64):
                sW, nW, elW, _ = run_sedov(ctx, torch, dist, rank, world, 256, mgs, args.steps, args.warmup, profile=False, carry=(args.rk2_mode == "carry"))
                out["weak_256_per_gpu"] = {"value": nW[0] * nW[1] * nW[2] * args.steps / elW / 1e6, "unit": "Mcell-updates/s", "ms_per_step": elW / args.steps * 1e3,
                                           "workload": f"{nW[0]}x{nW[1]}x{nW[2]}, 8 boxes of 128^3 per GPU"}
                del sW
                torch.cuda.empty_cache()
            # BASELINE config 5 on the same ranks: blast_amr_maxlev2.in, the SAME hierarchy on N GPUs (strong scaling)
            blk = run_amr(ctx, torch, dist, rank, world, 64 if share else 256, 4 if share else 30, 1 if share else 5)
            out["amr_maxlev2"] = compact(blk, keep=("value", "unit", "steps", "warmup", "ms_per_step", "scaling", "config"))
            torch.cuda.empty_cache()
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_ncell, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline_shell(ncell: int = 32, steps: int = 2):
    """the oracle's RadhydroShell (same deck, a 32^3 sample: one hydro RK2 update + 10 radiation substeps per step) on the host cores"""
    import numpy as np
    from oracle.pyoracle import SHELL, Oracle
    from quokka_amd.radhydro import ShellConstants
    threads, note = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(threads)
    o = Oracle("direct")
    tab = np.loadtxt(os.path.join(ROOT, "tests", "golden", "dust_shell_initial_conditions.txt"), skiprows=1)
    L = ShellConstants.L_box
    s = o.sim(SHELL, 3, [ncell] * 3, [0, 0, 0], [L] * 3, [1, 1, 1], max_grid_size=[16] * 3, table=(tab[:, 0], tab[:, 2], tab[:, 3]), rad_pow_mode=0)
    assert s.step()
    t0 = time.perf_counter()
    for _ in range(steps):
        assert s.step()
    el = time.perf_counter() - t0
    return {"value": ncell ** 3 * steps / el / 1e6, "unit": "Mcell-updates/s", "cores": threads, "kind": "port",
            "sample": f"RadhydroShell {ncell}^3 in 16^3 boxes, {steps} steps in {el:.1f} s ({note}); CPU restatement of the reference algorithm, "
                      "not the reference binary"}


if __name__ == "__main__":
    main()
