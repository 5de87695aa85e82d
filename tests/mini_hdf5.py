"""A reader of the few HDF5 structures the cloudy_cooling_tools table files use (version-0 superblock, one flat root group, version-1 object
headers, contiguous datasets, version-1 attributes), written from the HDF5 file-format specification for the tests: the second, independent
implementation the library's own reader (quokka_amd/csrc/qk_hdf5_mini.hpp) is compared with.  h5py is not installed in this image."""
import struct

import numpy as np


class MiniH5:
    def __init__(self, path):
        self.d = d = open(path, "rb").read()
        assert d[:8] == b"\x89HDF\r\n\x1a\n" and d[8] == 0 and d[13] == 8 and d[14] == 8, "not a version-0 HDF5 file with 8-byte offsets"
        btree, heap = struct.unpack_from("<QQ", d, 56 + 24)
        self.objects = self._entries(btree, heap)

    def _entries(self, btree, heap):
        d = self.d
        assert d[heap:heap + 4] == b"HEAP"
        seg = struct.unpack_from("<Q", d, heap + 24)[0]
        out = {}

        def node(a):
            assert d[a:a + 4] == b"TREE"
            _, level, used = struct.unpack_from("<BBH", d, a + 4)
            for n in range(used):
                child = struct.unpack_from("<Q", d, a + 24 + 16 * n + 8)[0]
                if level > 0:
                    node(child)
                    continue
                assert d[child:child + 4] == b"SNOD"
                for s in range(struct.unpack_from("<H", d, child + 6)[0]):
                    nameoff, hdr = struct.unpack_from("<QQ", d, child + 8 + 40 * s)
                    name = d[seg + nameoff:d.index(b"\0", seg + nameoff)].decode()
                    out[name] = hdr

        node(btree)
        return out

    def _messages(self, hdr):
        d = self.d
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", d, hdr)
        assert ver == 1
        blocks, msgs = [(hdr + 16, hsize)], []
        while blocks and len(msgs) < nmsg:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(msgs) < nmsg:
                t, s = struct.unpack_from("<HH", d, p)
                body = d[p + 8:p + 8 + s]
                if t == 0x10:
                    blocks.append(struct.unpack_from("<QQ", body, 0))
                msgs.append((t, body))
                p += 8 + s
        return msgs

    @staticmethod
    def _dims(b):
        off = 8 if b[0] == 1 else 4
        return [struct.unpack_from("<Q", b, off + 8 * i)[0] for i in range(b[1])]

    @staticmethod
    def _dtype(b):
        cls, bits, size = b[0] & 0xF, b[1], struct.unpack_from("<I", b, 4)[0]
        order = ">" if bits & 1 else "<"
        if cls == 1:
            return np.dtype(f"{order}f{size}")
        assert cls == 0
        return np.dtype(f"{order}{'i' if bits & 8 else 'u'}{size}")

    def dataset(self, name):
        """(values in file order as a native array, {attribute: array})"""
        dims = dt = data = None
        attrs = {}
        for t, b in self._messages(self.objects[name]):
            if t == 1:
                dims = self._dims(b)
            elif t == 3:
                dt = self._dtype(b)
            elif t == 8:
                assert b[0] == 3 and b[1] == 1, "contiguous version-3 layout only"
                addr, size = struct.unpack_from("<QQ", b, 2)
                data = self.d[addr:addr + size]
            elif t == 0xC:
                assert b[0] == 1
                nsz, dsz, ssz = struct.unpack_from("<HHH", b, 2)
                pad = lambda n: (n + 7) // 8 * 8
                p = 8
                aname = b[p:p + nsz].split(b"\0")[0].decode()
                p += pad(nsz)
                adt_raw = b[p:p + dsz]
                p += pad(dsz)
                adims = self._dims(b[p:p + ssz])
                p += pad(ssz)
                if (adt_raw[0] & 0xF) in (0, 1):
                    adt = self._dtype(adt_raw)
                    n = int(np.prod(adims)) if adims else 1
                    attrs[aname] = np.frombuffer(b[p:p + n * adt.itemsize], dtype=adt).astype(adt.newbyteorder("="))
        arr = np.frombuffer(data, dtype=dt).reshape(dims)
        return arr.astype(dt.newbyteorder("=")), attrs


def cloudy_file_arrays(path):
    """Parameter1[n0], Temperature[n1], Cooling / Heating / MMW [n0][n1] of a cloudy_cooling_tools file, as H5Dread would deliver them"""
    h = MiniH5(path)
    out = {k: np.ascontiguousarray(h.dataset(k)[0], dtype=np.float64) for k in ("Parameter1", "Temperature", "Cooling", "Heating", "MMW")}
    _, attrs = h.dataset("Cooling")
    assert int(attrs["Rank"][0]) == 2 and [int(v) for v in attrs["Dimension"]] == list(out["Cooling"].shape)
    return out
