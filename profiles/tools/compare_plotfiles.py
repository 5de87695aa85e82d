import sys, numpy as np
sys.path.insert(0, "/root/repo")
from quokka_amd import plotfile as pf
a, b = sys.argv[1], sys.argv[2]
A, B = pf.read_plotfile(a), pf.read_plotfile(b)
print(a, "levels", A.finest_level, B.finest_level)
for l, (la, lb) in enumerate(zip(A.levels, B.levels)):
    same = la.boxes == lb.boxes
    print(" level", l, "boxes", len(la.boxes), "same grids", same, la.boxes[:6])
    if not same:
        continue
    for n, v in enumerate(A.varnames):
        worst, where = 0.0, None
        for bi, (fa, fb) in enumerate(zip(la.fabs, lb.fabs)):
            d = np.abs(fa[n] - fb[n])
            if d.max() > worst:
                worst = float(d.max()); where = (bi, np.unravel_index(d.argmax(), d.shape), float(np.abs(fa[n]).max()))
        if worst > 0:
            cells = []
            for bi, (fa, fb) in enumerate(zip(la.fabs, lb.fabs)):
                lo = la.boxes[bi][0]
                for idx in np.argwhere(fa[n] != fb[n]):
                    cells.append((int(idx[2]) + lo[0], int(idx[1]) + lo[1], int(idx[0]) + lo[2]))
            c = np.array(cells)
            print("   ", v, worst, where, "differing cells:", len(cells), "bounding box", c.min(axis=0).tolist(), c.max(axis=0).tolist())
