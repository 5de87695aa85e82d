#!/bin/bash
# kernel-trace of config 5 through the C++ host, ordinary vs speculative coarse step: span, union-busy and summed kernel time over the evolve,
# per-queue busy time, the biggest idle gaps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=$R/quokka_amd/host
for o in 0 1; do
  O=$R/gpurun_out/cxx_amr_conc_$o; rm -rf $O; mkdir -p $O
  (cd $B && rocprofv3 --kernel-trace --output-format csv -d $O/kt -- $B/bin/ref_HydroBlast3D $B/decks/blast_amr_maxlev2.in max_timesteps=55 hydro.rk2_carry_rhs=1 plotfile_interval=-1 checkpoint_interval=-1 qk.overlap_children=$o > $O/run.log 2>&1)
  grep -E "figure-of-merit|speculative" $O/run.log | tr '\n' ' '; echo
  python - "$O" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/kt/**/*kernel_trace.csv", recursive=True)[0]
ks = []
for r in csv.DictReader(open(f)):
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")))
ks.sort()
m = [i for i, k in enumerate(ks) if "k_sweep_march" in k[2]]
ks = ks[m[0]:m[-1] + 1]
span = (max(k[1] for k in ks) - ks[0][0]) / 1e6
tot = sum(k[1] - k[0] for k in ks) / 1e6
busy, end = 0, 0
gaps = []
for s, e, n, q in ks:
    if s > end:
        if end: gaps.append(((s - end) / 1e3, n))
        busy += e - s; end = e
    elif e > end:
        busy += e - end; end = e
perq = collections.Counter()
for s, e, n, q in ks: perq[q] += (e - s) / 1e6
print(f"  evolve span {span:.1f} ms, union busy {busy/1e6:.1f} ms, summed kernels {tot:.1f} ms, per queue {dict((k, round(v,1)) for k,v in perq.items())}, gaps > 15 us: {sum(1 for g in gaps if g[0] > 15)} totalling {sum(g[0] for g in gaps if g[0] > 15)/1e3:.1f} ms")
PY
  rm -rf $O/kt
done
