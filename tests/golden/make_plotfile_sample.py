"""Writes tests/golden/plotfile_sample/: a 4x2x2-cell, 2-box, 2-component single-level plotfile with the writer of
quokka_amd/plotfile.py (layout restated from the reference's header writers, src/io/DiagFramePlane.cpp:321-386,517-572,691-699;
not produced by AMReX).  `python tests/golden/make_plotfile_sample.py` regenerates it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def write(name):
    from quokka_amd import plotfile
    boxes = [([0, 0, 0], [1, 1, 1]), ([2, 0, 0], [3, 1, 1])]
    os.makedirs(os.path.join(name, "Level_0"))
    plotfile.write_plotfile_header(name, ["gasDensity", "gasEnergy"], 3, 0.125, [0.0, 0.0, 0.0], [2.0, 1.0, 1.0], [([0, 0, 0], [3, 1, 1])], [3], [[0.5, 0.5, 0.5]],
                                   [boxes])
    fabs = [np.arange(16, dtype=np.float64).reshape(2, 2, 2, 2) + 100.0 * b for b in range(2)]
    plotfile.write_vismf(os.path.join(name, "Level_0", "Cell"), boxes, [0, 0], 0, fabs, 2, 0, 3)


if __name__ == "__main__":
    import shutil
    out = os.path.join(ROOT, "tests", "golden", "plotfile_sample")
    shutil.rmtree(out, ignore_errors=True)
    write(out)
