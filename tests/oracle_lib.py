"""ctypes access to the CPU oracle (oracle/liboracle.so) for the tests.

The oracle is the checker: tests compare the HIP path with it.  It is never
imported by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref.so")
NBB = 403
NCELLS = NBB ** 3


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), ORACLE_SO])
        L = ctypes.CDLL(ORACLE_SO)
        c = ctypes
        L.orc_parse_float_bits.restype = c.c_uint32
        L.orc_parse_float_bits.argtypes = [c.c_char_p, c.POINTER(c.c_char_p), c.POINTER(c.c_int)]
        L.orc_parse_buffer.restype = c.c_int64
        L.orc_parse_buffer.argtypes = [c.c_char_p, c.c_uint64, c.c_void_p, c.c_uint64]
        L.orc_cell_index.restype = c.c_int
        L.orc_cell_index.argtypes = [c.c_float]
        L.orc_cell_keys.restype = None
        L.orc_cell_keys.argtypes = [c.c_void_p, c.c_uint64, c.c_void_p]
        L.orc_dist2.restype = c.c_float
        L.orc_dist2.argtypes = [c.c_void_p, c.c_void_p]
        L.orc_within.restype = c.c_int
        L.orc_within.argtypes = [c.c_float]
        L.orc_count_pairs.restype = c.c_int64
        L.orc_count_pairs.argtypes = [c.c_void_p, c.c_uint64, c.c_int, c.c_int, c.c_int,
                                      c.POINTER(c.c_int64), c.c_void_p, c.c_void_p]
        L.orc_count_pairs_bruteforce.restype = c.c_int64
        L.orc_count_pairs_bruteforce.argtypes = [c.c_void_p, c.c_uint64]
        L.orc_count_pairs_anycell.restype = c.c_int64
        L.orc_count_pairs_anycell.argtypes = [c.c_void_p, c.c_uint64]
        L.orc_set_geometry.restype = None
        L.orc_set_geometry.argtypes = [c.c_double, c.c_double, c.c_int, c.c_float]
        self.L = L

    def parse_token(self, tok):
        err = ctypes.c_int()
        buf = ctypes.create_string_buffer(tok.encode() + b"\n")
        end = ctypes.c_char_p()
        bits = self.L.orc_parse_float_bits(buf, None, ctypes.byref(err))
        return bits, err.value

    def parse_text(self, data):
        """bytes of a .xyz file -> (n,3) float32"""
        buf = ctypes.create_string_buffer(data + b"\0")
        cap = len(data) // 6 + 16
        out = np.empty((cap, 3), dtype=np.float32)
        n = self.L.orc_parse_buffer(buf, len(data), out.ctypes.data, cap)
        if n < 0:
            raise ValueError("oracle tokenizer error")
        return out[:n].copy()

    def cell_keys(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        keys = np.empty(xyz.shape[0], dtype=np.int64)
        self.L.orc_cell_keys(xyz.ctypes.data, xyz.shape[0], keys.ctypes.data)
        return keys

    def cell_index(self, f):
        return self.L.orc_cell_index(ctypes.c_float(f))

    def dist2_bits(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        d = np.float32(self.L.orc_dist2(a.ctypes.data, b.ctypes.data))
        return int(np.array([d], dtype=np.float32).view(np.uint32)[0]), bool(self.L.orc_within(ctypes.c_float(d)))

    def count_pairs(self, xyz, threads=8, x_lo=0, x_hi=403, want_sorted=False):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        n = xyz.shape[0]
        tests = ctypes.c_int64()
        srt = np.empty((max(n, 1), 3), dtype=np.float32) if want_sorted else None
        start = np.empty(NCELLS + 1, dtype=np.uint32) if want_sorted else None
        r = self.L.orc_count_pairs(xyz.ctypes.data, n, threads, x_lo, x_hi, ctypes.byref(tests),
                                   srt.ctypes.data if want_sorted else None,
                                   start.ctypes.data if want_sorted else None)
        if r < 0:
            raise ValueError("oracle count error %d" % r)
        if want_sorted:
            return int(r), int(tests.value), srt[:n], start
        return int(r), int(tests.value)

    def set_geometry(self, origin=-10.0, cell=0.05, ncell=400, d2max=0.0025):
        """Other constants for the same expressions (process-wide; call with no arguments to restore the reference's)."""
        self.L.orc_set_geometry(origin, cell, ncell, d2max)

    def brute_anycell(self, xyz):
        """All pairs within the cutoff, no grid involved (O(n^2))."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        return int(self.L.orc_count_pairs_anycell(xyz.ctypes.data, xyz.shape[0]))

    def brute(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        return int(self.L.orc_count_pairs_bruteforce(xyz.ctypes.data, xyz.shape[0]))


def slab_points(xyz, keys, x_lo, x_hi):
    """Points a rank owning home layers [x_lo,x_hi) must hold: its own plus layer x_hi."""
    cx = keys // (NBB * NBB)
    return xyz[(cx >= x_lo) & (cx <= x_hi)]
