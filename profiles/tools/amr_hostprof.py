"""Host-side profile of the AMR driver (where do the ~2 ms per small-level step go?): python profiles/tools/amr_hostprof.py [nsteps]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.getcwd())
import torch

from quokka_amd.amr_simulation import sedov_amr_problem
from quokka_amd.multifab import Context

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
amr = sedov_amr_problem(Context(0), 256, 2, max_grid_size=128, blocking_factor=32)
for _ in range(5):
    amr.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(nsteps):
    amr.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
