#!/bin/bash
# config 5 through the C++ host (the reference's unchanged HydroBlast3D on blast_amr_maxlev2.in, 55 coarse steps): ordinary order against the
# speculative coarse step (children beside the far boxes, deferred verdicts), and the Python host beside them
cd quokka_amd/host
for rep in 1 2; do
  for o in 0 1; do
    ./bin/ref_HydroBlast3D decks/blast_amr_maxlev2.in max_timesteps=55 hydro.rk2_carry_rhs=1 plotfile_interval=-1 checkpoint_interval=-1 qk.overlap_children=$o 2>&1 | grep -E "figure-of-merit|speculative|Energy conservation" | tr '\n' ' ' | sed "s/^/cxx overlap_children=$o rep $rep: /"; echo
  done
done
cd ../..
python bench.py --workload amr --steps 50 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('python host: %.1f M' % d['value'])"
