"""two qkhost::dumpFabs dumps (QK_DUMP_BEFORE_SOURCE / QK_DUMP_COARSE_FOR_INTERP): cells of the fabs present in BOTH that differ, per component.
usage: compare_fab_dumps.py <prefix A> <nranks A> <prefix B> <nranks B> [ghost width 4]"""
import sys
import numpy as np


def load(prefix, nranks):
    out = {}
    for r in range(nranks):
        lines = open(f"{prefix}.rank{r}.txt").read().strip().split("\n")
        nc = int(lines[0].split()[-1])
        data = np.fromfile(f"{prefix}.rank{r}.bin")
        off = 0
        for ln in lines[1:]:
            v = [int(x) for x in ln.split()]
            shape = (v[5] - v[2] + 1, v[4] - v[1] + 1, v[3] - v[0] + 1)
            n = nc * int(np.prod(shape))
            out[(tuple(v[:3]), tuple(v[3:]))] = data[off:off + n].reshape((nc,) + shape)
            off += n
        print(prefix, "rank", r, lines[0], "fabs", len(lines) - 1)
    return out


A, B = load(sys.argv[1], int(sys.argv[2])), load(sys.argv[3], int(sys.argv[4]))
ng = int(sys.argv[5]) if len(sys.argv) > 5 else 4
print("fabs only in A:", sorted(set(A) - set(B)), " only in B:", sorted(set(B) - set(A)))
for key in sorted(set(A) & set(B)):
    a, b = A[key], B[key]
    for n in range(a.shape[0]):
        bad = ~((a[n] == b[n]) | (np.isnan(a[n]) & np.isnan(b[n])))
        if bad.any():
            idx = np.argwhere(bad)
            lo = key[0]
            cells = [(int(i[2]) + lo[0], int(i[1]) + lo[1]) for i in idx]
            valid = sum(1 for i in idx if ng <= i[2] < a.shape[3] - ng and ng <= i[1] < a.shape[2] - ng)
            print("fab", key, "comp", n, "differing cells", len(cells), "(valid:", valid, ") x", min(c[0] for c in cells), max(c[0] for c in cells), "y", min(c[1] for c in cells),
                  max(c[1] for c in cells), "first", cells[:4], "A", a[n][bad][:3], "B", b[n][bad][:3])
print("done")
