"""On-disk layout against expectations TYPED BY HAND from the reference's own header writers — not produced by the writer under test
(tests/test_plotfile.py compares the writer with itself).  The only statements of the AMReX plotfile / VisMF layout inside the reference tree
are its 2-D slice writer, src/io/DiagFramePlane.cpp:

  Write2DPlotfileHeader  :321-386   the `Header` file (the same sequence as amrex::WriteGenericPlotfileHeader, one dimension lower)
  write_2D_header        :654-662   the text line in front of every fab in Cell_D_*
  Find2FOffsets          :642-646   FabOnDisk offsets: fab header bytes + numPts * ncomp * 8, fabs of one rank back to back
  Write2DMFHeader        :517-572   `Cell_H`: version, how, ncomp, ngrow, box list, FabOnDisk list, min / max tables (scientific, precision 16)

Each expected string below follows those statements line by line for a 2-D level of two 8 x 8 boxes.  Two pieces come from AMReX itself (absent
from the tree) and are quoted from its file format as every plotfile on disk shows it: the native real descriptor printed after "FAB ", and
`operator<<` of the FabOnDisk vector ("<n>\\nFabOnDisk: <file> <offset>\\n...")."""
import os

import numpy as np

from quokka_amd import plotfile

BOXES = [([0, 0, 0], [7, 7, 0]), ([8, 0, 0], [15, 7, 0])]

# Write2DPlotfileHeader, statement by statement (precision(17), general format: 0.5 -> "0.5", 1.0 -> "1", 0.0625 -> "0.0625")
HEADER = (
    "HyperCLaw-V1.1\n"          # :331 versionName
    "2\n"                       # :332 varnames.size()
    "gasDensity\n"              # :333-335
    "gasEnergy\n"
    "2\n"                       # :336 lowerSpaceDim
    "0.5\n"                     # :337 time
    "0\n"                       # :338 finest_level
    "0 0 \n"                    # :339-342 ProbLo(idim) << ' ', then '\n'
    "1 0.5 \n"                  # :343-346 ProbHi
    "\n"                        # :347-350 ref_ratio of levels 0 .. finest-1: none
    "((0,0) (15,7) (0,0)) \n"   # :351-355 printLowerDimBox(Domain) << ' '
    "7 \n"                      # :356-359 level_steps
    "0.0625 0.0625 \n"          # :360-365 CellSizeArray
    "0\n"                       # :366 Coord (cartesian)
    "0\n"                       # :367
    "0 2 0.5\n"                 # :370 level, number of grids, time
    "7\n"                       # :371 level_steps[level]
    "0 0.5\n"                   # :379-381 RealBox of grid 0: lo hi per dimension
    "0 0.5\n"
    "0.5 1\n"                   # grid 1
    "0 0.5\n"
    "Level_0/Cell\n"            # :383 MultiFabHeaderPath(level, "Level_", "Cell")
)

REAL = "((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))"  # amrex::FPC::NativeRealDescriptor(): IEEE double, little endian
FAB0 = "FAB " + REAL + "((0,0) (7,7) (0,0)) 2\n"                 # write_2D_header :656-661
FAB1 = "FAB " + REAL + "((8,0) (15,7) (0,0)) 2\n"
OFFSET1 = len(FAB0) + 8 * 8 * 2 * 8                               # Find2FOffsets :644-646: header bytes + numPts * nComps * 8
assert (len(FAB0), OFFSET1) == (80, 1104)                         # (counted by hand: 4 + 54 + 19 + 3)

# Write2DMFHeader, statement by statement (`slice_layout`: the blank after the closing parenthesis of the box list, :552)
CELL_H = (
    "1\n"                        # :540 m_vers (Version_v1)
    "1\n"                        # :541 m_how (NFiles)
    "2\n"                        # :542 m_ncomp
    "0\n"                        # :543-547 m_ngrow, all equal
    "(2 0\n"                     # :549 '(' << size << " 0"
    "((0,0) (7,7) (0,0))\n"      # :550-553
    "((8,0) (15,7) (0,0))\n"
    ") \n"                       # :554
    "2\n"                        # :556 m_fod: the vector's size,
    "FabOnDisk: Cell_D_00000 0\n"      # one line per fab,
    "FabOnDisk: Cell_D_00000 1104\n"
    "\n"                         # and the '\n' of :556
    "2,2\n"                      # :558 m_min.size() "," m_min[0].size()
    "1.0000000000000000e+00,-3.0000000000000000e+00,\n"   # :559-565 scientific, precision(16), "," after every value
    "2.0000000000000000e+00,2.5000000000000000e-01,\n"
    "\n"                         # :567
    "2,2\n"                      # :569
    "1.0000000000000000e+00,-3.0000000000000000e+00,\n"   # :570-575
    "2.0000000000000000e+00,2.5000000000000000e-01,\n"
)


def _fabs():
    a = np.empty((2, 1, 8, 8))
    a[0], a[1] = 1.0, -3.0
    b = np.empty((2, 1, 8, 8))
    b[0], b[1] = 2.0, 0.25
    return [a, b]


def test_header_bytes_equal_the_hand_typed_expectation(tmp_path):
    name = str(tmp_path / "plt00007")
    plotfile._prebuild(name, 1, 0)
    plotfile.write_plotfile_header(name, ["gasDensity", "gasEnergy"], 2, 0.5, [0.0, 0.0], [1.0, 0.5], [([0, 0, 0], [15, 7, 0])], [7], [[0.0625, 0.0625]], [BOXES])
    assert open(os.path.join(name, "Header")).read() == HEADER


def test_vismf_bytes_equal_the_hand_typed_expectation(tmp_path):
    prefix = str(tmp_path / "Cell")
    fabs = _fabs()
    plotfile.write_vismf(prefix, BOXES, [0, 0], 0, fabs, 2, 0, 2, slice_layout=True)
    assert open(prefix + "_H").read() == CELL_H
    raw = open(str(tmp_path / "Cell_D_00000"), "rb").read()
    assert len(raw) == 2 * OFFSET1 + (len(FAB1) - len(FAB0))
    assert raw[:len(FAB0)] == FAB0.encode()
    assert raw[OFFSET1:OFFSET1 + len(FAB1)] == FAB1.encode()
    # the payload: components outermost, x fastest, little-endian IEEE doubles (amrex::FArrayBox::dataPtr() written as it lies, :471 / :497)
    assert np.array_equal(np.frombuffer(raw[len(FAB0):OFFSET1], dtype="<f8"), np.concatenate([np.full(64, 1.0), np.full(64, -3.0)]))
    assert np.array_equal(np.frombuffer(raw[OFFSET1 + len(FAB1):], dtype="<f8"), np.concatenate([np.full(64, 2.0), np.full(64, 0.25)]))
    # AMReX's own VisMF header (what plotfiles and checkpoints of the evolve loop carry) closes the box list without the blank: the reader takes both
    plotfile.write_vismf(prefix, BOXES, [0, 0], 0, fabs, 2, 0, 2)
    assert open(prefix + "_H").read() == CELL_H.replace(") \n", ")\n")
    r = plotfile.read_vismf(prefix)
    assert r.boxes == BOXES and np.array_equal(r.fabs[1], fabs[1]) and np.array_equal(r.minima, [[1.0, -3.0], [2.0, 0.25]])
