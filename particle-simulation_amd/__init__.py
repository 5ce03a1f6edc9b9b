"""MI355X-native pair counting: Python mirror of the reference's seams.

The reference (Remmarg0rPA/particle-simulation) is a C program; its hot path is
reached through init_hashmap / insert / flatten / count (src/hashmap.h:31-34,
src/main.c:236).  This package binds the C ABI of libpaircount.so
(include/paircount.h), which replaces exactly those calls, and exposes

    PairCounter      -- create / append / finish, the streaming seam
    DeviceStages     -- the stateless device-pointer stages on torch tensors

PyTorch is used only as the owner of device memory and streams.  There is no
CPU fallback: importing works anywhere, but every compute call needs the HIP
library and a GPU and raises PairCountError otherwise.

The directory name contains a hyphen (it follows the reference repository's
name); load it with `load_package()` from the repository root's conftest or via
importlib, under the module name ``particle_simulation_amd``.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PAIRCOUNT_LIB") or os.path.join(HERE, "libpaircount.so")   # (override: A/B runs of library builds)

PC_NBB = 403
PC_NCELLS = PC_NBB ** 3
PC_NLAYERS = 400
TABLE_LEN = PC_NCELLS + 2

ERR_NAMES = {0: "PC_OK", -1: "PC_ERR_ARG", -2: "PC_ERR_HIP", -3: "PC_ERR_DOMAIN", -4: "PC_ERR_NOMEM", -5: "PC_ERR_STATE", -6: "PC_ERR_TEXT"}

# every symbol include/paircount.h declares: (name, restype, argtypes)
_c = ctypes
_vp, _u64, _i32, _sz = _c.c_void_p, _c.c_uint64, _c.c_int, _c.c_size_t
ABI = [
    ("pc_last_error", _c.c_char_p, []),
    ("pc_version", _c.c_char_p, []),
    ("pc_create", _i32, [_c.POINTER(_vp), _i32, _u64]),
    ("pc_set_slab", _i32, [_vp, _i32, _i32]),
    ("pc_append", _i32, [_vp, _vp, _u64]),
    ("pc_host_alloc", _i32, [_c.POINTER(_vp), _u64]),
    ("pc_host_free", _i32, [_vp]),
    ("pc_append_device", _i32, [_vp, _vp, _u64, _vp]),
    ("pc_append_text", _i32, [_vp, _c.c_char_p, _u64, _c.POINTER(_u64)]),
    ("pc_parse_workspace_bytes", _sz, [_u64]),
    ("pc_parse_text_device", _i32, [_vp, _u64, _vp, _u64, _vp, _vp, _vp, _sz, _vp]),
    ("pc_export_layers", _i32, [_vp, _c.POINTER(_vp), _c.POINTER(_c.c_uint32)]),
    ("pc_import_device", _i32, [_vp, _i32, _vp, _u64]),
    ("pc_finish", _i32, [_vp, _c.POINTER(_c.c_int64)]),
    ("pc_tests", _i32, [_vp, _c.POINTER(_c.c_int64)]),
    ("pc_size", _i32, [_vp, _c.POINTER(_u64)]),
    ("pc_phase_ms", _i32, [_vp, _c.POINTER(_c.c_float)]),
    ("pc_destroy", None, [_vp]),
    ("pc_workspace_bytes", _sz, [_u64]),
    ("pc_cell_keys_device", _i32, [_vp, _u64, _vp, _vp, _vp]),
    ("pc_bin_device", _i32, [_vp, _u64, _vp, _vp, _vp, _sz, _vp, _vp]),
    ("pc_count_sorted_device", _i32, [_vp, _vp, _u64, _i32, _i32, _vp, _vp]),
    ("pc_count_tests_device", _i32, [_vp, _i32, _i32, _vp, _vp]),
    ("pc_count_pairs_device", _i32, [_vp, _u64, _i32, _i32, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    ("pc_noalloc_prepare", _i32, []),
    ("pc_count_pairs_noalloc", _i32, [_vp, _u64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    ("pc_geometry_default", _i32, [_vp]),
    ("pc_set_geometry", _i32, [_vp, _vp]),
    ("pc_cell_keys_geometry_device", _i32, [_vp, _vp, _u64, _vp, _vp, _vp]),
    ("pc_count_pairs_geometry_device", _i32, [_vp, _vp, _u64, _i32, _i32, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    ("pc_bin_geometry_device", _i32, [_vp, _vp, _u64, _vp, _vp, _vp, _sz, _vp, _vp]),
    ("pc_selfcheck_cell_index_geometry_device", _i32, [_vp, _vp, _vp]),
    ("pc_set_variant", _i32, [_c.c_char_p, _i32]),
    ("pc_context_set_variant", _i32, [_vp, _c.c_char_p, _i32]),
    ("pc_profile_enable", _i32, [_i32]),
    ("pc_profile_read", _i32, [_c.c_char_p, _sz]),
    ("pc_selfcheck_cell_index_device", _i32, [_vp, _vp]),
    ("pc_generate_device", _i32, [_i32, _u64, _c.c_int64, _u64, _u64, _vp, _vp]),
]


class Geometry(ctypes.Structure):
    """pc_geometry of include/paircount.h: domain origin, cell edge, cells per axis, squared cutoff."""
    _fields_ = [("origin", ctypes.c_double), ("cell", ctypes.c_double), ("ncell", ctypes.c_int32), ("d2max", ctypes.c_float)]

    def __init__(self, origin=-10.0, cell=0.05, ncell=400, d2max=None, cutoff=None):
        import numpy as np
        if d2max is None:
            d2max = 0.0025 if cutoff is None else float(np.float32(cutoff) * np.float32(cutoff))
        super().__init__(float(origin), float(cell), int(ncell), float(np.float32(d2max)))


class PairCountError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, str(code)), message))
        self.code = code


_lib = None


def lib():
    """The loaded libpaircount.so.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PairCountError(-2, "HIP library %s is missing; run __graft_entry__.build()" % LIB_PATH)
        # One HIP runtime per process: PyTorch ships its own libamdhip64 and, loaded second, finds the GPUs taken
        # ("No HIP GPUs are available").  Loaded first, its copy also satisfies this library's dependency.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in ABI:
            fn = getattr(handle, name)      # AttributeError if a declared symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def _check(rc):
    if rc != 0:
        raise PairCountError(rc, lib().pc_last_error().decode())


_variants = {}          # what set_variant() was last given, per name (the library has no getter)


def set_variant(name, value):
    _check(lib().pc_set_variant(name.encode(), int(value)))
    _variants[name] = int(value)


def profile_enable(on=True):
    """on: False/0 off, True/1 every launch, k > 1 every k-th launch of each kernel."""
    _check(lib().pc_profile_enable(int(on)))


def profile_read():
    """{kernel: (timed_launches, launches, total_ms)} since the last profile_enable(True)."""
    buf = ctypes.create_string_buffer(8192)
    _check(lib().pc_profile_read(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, timed, launches, ms = line.split()
        out[name] = (int(timed), int(launches), float(ms))
    return out


def _host_ptr(arr):
    import numpy as np
    a = np.ascontiguousarray(arr, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 3:
        a = a.reshape(-1, 3)
    return a, a.ctypes.data_as(ctypes.c_void_p), a.shape[0]


class HostBlock:
    """A page-locked host buffer of `points` float triples (pc_host_alloc); `.array` is its numpy view."""

    def __init__(self, points):
        import numpy as np
        self._p = ctypes.c_void_p()
        _check(lib().pc_host_alloc(ctypes.byref(self._p), int(points) * 12))
        self.array = np.ctypeslib.as_array(ctypes.cast(self._p, ctypes.POINTER(ctypes.c_float)), shape=(int(points), 3))

    def close(self):
        if self._p:
            self.array = None
            _check(lib().pc_host_free(self._p))
            self._p = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class PairCounter:
    """Streaming seam: the object a host program holds instead of a Hashmap.

    pc = PairCounter(device=0, expected_points=n)   # init_hashmap   (main.c:316-317)
    pc.append(xyz_block)                            # insert, bulk   (main.c:137); thread-safe
    npairs = pc.finish()                            # flatten+count  (main.c:331-335)
    """

    def __init__(self, device=0, expected_points=0, slab=None, geometry=None):
        self._h = ctypes.c_void_p()
        _check(lib().pc_create(ctypes.byref(self._h), int(device), int(expected_points)))
        if slab is not None:
            _check(lib().pc_set_slab(self._h, int(slab[0]), int(slab[1])))
        if geometry is not None:
            _check(lib().pc_set_geometry(self._h, ctypes.byref(geometry)))

    def set_variant(self, name, value):
        """This context's own "bin" / "count" variant (pc_context_set_variant)."""
        _check(lib().pc_context_set_variant(self._h, name.encode(), int(value)))

    def append(self, xyz):
        a, p, n = _host_ptr(xyz)
        _check(lib().pc_append(self._h, p, n))

    def append_device(self, t):
        import torch
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        n = t.numel() // 3
        _check(lib().pc_append_device(self._h, t.data_ptr(), n, torch.cuda.current_stream(t.device).cuda_stream))

    def append_text(self, data):
        """A block of "x y z" lines (bytes, whole records): tokenized on the GPU.  Returns the points appended."""
        n = ctypes.c_uint64()
        _check(lib().pc_append_text(self._h, data, len(data), ctypes.byref(n)))
        return n.value

    def export_layers(self):
        """Groups the context's points by x layer on its device.  Returns (device address of the array,
        numpy array of 404 layer starts); see pc_export_layers."""
        import numpy as np
        ptr = ctypes.c_void_p()
        starts = (ctypes.c_uint32 * (PC_NBB + 1))()
        _check(lib().pc_export_layers(self._h, ctypes.byref(ptr), starts))
        return ptr.value, np.frombuffer(starts, dtype=np.uint32).astype(np.int64)

    def import_device(self, src_device, address, n):
        """n points at a device address of GPU `src_device` (e.g. a layer range of export_layers) into this context."""
        _check(lib().pc_import_device(self._h, int(src_device), ctypes.c_void_p(int(address)), int(n)))

    def size(self):
        n = ctypes.c_uint64()
        _check(lib().pc_size(self._h, ctypes.byref(n)))
        return n.value

    def finish(self):
        r = ctypes.c_int64()
        _check(lib().pc_finish(self._h, ctypes.byref(r)))
        return r.value

    def tests(self):
        r = ctypes.c_int64()
        _check(lib().pc_tests(self._h, ctypes.byref(r)))
        return r.value

    def phase_ms(self):
        ms = (ctypes.c_float * 2)()
        _check(lib().pc_phase_ms(self._h, ms))
        return {"binning": ms[0], "count": ms[1]}

    def close(self):
        if self._h:
            lib().pc_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceStages:
    """The stateless stages on torch CUDA tensors (torch owns memory + stream).

    Buffers are allocated once for `capacity` points and reused; no call here
    synchronises or allocates, so the sequence can be timed with events on the
    current torch stream.

    Under set_variant("count", 7) count_pairs() may leave the sorted array and the
    table unwritten (the library then runs its last binning level inside the pair
    test); `.sorted`, `.table`, count_sorted() and count_tests() bin the same points
    again first when that may have happened, so they always see pc_bin_device's results.
    """

    def __init__(self, capacity, device="cuda:0"):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.capacity = int(capacity)
        n = max(self.capacity, 1)
        self.ws_bytes = lib().pc_workspace_bytes(n)
        with torch.cuda.device(self.device):
            self._sorted = torch.empty((n, 3), dtype=torch.float32, device=self.device)
            self._table = torch.empty(TABLE_LEN, dtype=torch.int32, device=self.device)
            self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
            self.result = torch.zeros(2, dtype=torch.int64, device=self.device)   # [pairs, tests]
            self.status = torch.zeros(1, dtype=torch.int32, device=self.device)

        self._unbinned = None     # (xyz, geometry) of a count_pairs() whose table may not have been written

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def _binned(self):
        if self._unbinned is not None:
            xyz, geometry = self._unbinned
            self.bin(xyz, geometry=geometry)

    @property
    def sorted(self):
        self._binned()
        return self._sorted

    @property
    def table(self):
        self._binned()
        return self._table

    def _n(self, xyz):
        t = self.torch
        assert xyz.is_cuda and xyz.dtype == t.float32 and xyz.is_contiguous(), "xyz must be a contiguous float32 CUDA tensor"
        n = xyz.numel() // 3
        if n > self.capacity:
            raise PairCountError(-1, "n=%d exceeds DeviceStages capacity %d" % (n, self.capacity))
        return n

    def cell_keys(self, xyz, keys=None, geometry=None):
        n = self._n(xyz)
        if keys is None:
            keys = self.torch.empty(max(n, 1), dtype=self.torch.int32, device=self.device)
        if geometry is None:
            _check(lib().pc_cell_keys_device(xyz.data_ptr(), n, keys.data_ptr(), self.status.data_ptr(), self._stream()))
        else:
            _check(lib().pc_cell_keys_geometry_device(ctypes.byref(geometry), xyz.data_ptr(), n, keys.data_ptr(),
                                                      self.status.data_ptr(), self._stream()))
        return keys[:n]

    def bin(self, xyz, geometry=None):
        n = self._n(xyz)
        self._unbinned = None
        if geometry is None:
            _check(lib().pc_bin_device(xyz.data_ptr(), n, self._sorted.data_ptr(), self._table.data_ptr(),
                                       self.ws.data_ptr(), self.ws_bytes, self.status.data_ptr(), self._stream()))
        else:
            _check(lib().pc_bin_geometry_device(ctypes.byref(geometry), xyz.data_ptr(), n, self._sorted.data_ptr(),
                                                self._table.data_ptr(), self.ws.data_ptr(), self.ws_bytes,
                                                self.status.data_ptr(), self._stream()))
        return n

    def count_sorted(self, n, x_lo=0, x_hi=PC_NLAYERS, zero=True):
        if zero:
            self.result[0:1].zero_()
        _check(lib().pc_count_sorted_device(self.sorted.data_ptr(), self.table.data_ptr(), n, x_lo, x_hi,
                                            self.result.data_ptr(), self._stream()))

    def count_tests(self, x_lo=0, x_hi=PC_NLAYERS, zero=True):
        if zero:
            self.result[1:2].zero_()
        _check(lib().pc_count_tests_device(self.table.data_ptr(), x_lo, x_hi,
                                           self.result.data_ptr() + 8, self._stream()))

    def count_pairs(self, xyz, x_lo=0, x_hi=PC_NLAYERS, geometry=None):
        """Whole hot path, asynchronous; the count lands in self.result[0]."""
        n = self._n(xyz)
        if geometry is None:
            _check(lib().pc_count_pairs_device(xyz.data_ptr(), n, x_lo, x_hi, self._sorted.data_ptr(),
                                               self._table.data_ptr(), self.ws.data_ptr(), self.ws_bytes,
                                               self.result.data_ptr(), self.status.data_ptr(), self._stream()))
        else:
            _check(lib().pc_count_pairs_geometry_device(ctypes.byref(geometry), xyz.data_ptr(), n, x_lo, x_hi,
                                                        self._sorted.data_ptr(), self._table.data_ptr(), self.ws.data_ptr(),
                                                        self.ws_bytes, self.result.data_ptr(), self.status.data_ptr(),
                                                        self._stream()))
        self._unbinned = (xyz, geometry) if _variants.get("count") == 7 else None
        return n

    def pairs(self):
        """Synchronises and returns the last count; raises on a domain error."""
        st = int(self.status.item())
        if st & 1:
            self.status.zero_()
            raise PairCountError(-3, "a coordinate lies outside the reference's domain (-10,10)")
        return int(self.result[0].item())

    def tests(self):
        return int(self.result[1].item())


def generate_device(kind, seed, count, sigma_micro=0, first=0, device="cuda:0", x_lo=0, x_hi=PC_NLAYERS - 1,
                    piece=50_000_000):
    """Points [first, first+count) of the host generator's stream (hosttools.generate), made on the GPU
    bit for bit; with x_lo/x_hi only those whose x cell lies in [x_lo, x_hi], in stream order.
    A selection is made piece by piece, so only the kept points and one piece are ever resident."""
    import torch
    dev = torch.device(device)
    count, first = int(count), int(first)
    select = x_lo > 0 or x_hi < PC_NLAYERS - 1

    def make(start, m):
        out = torch.empty((m, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _check(lib().pc_generate_device(int(kind), int(seed), int(sigma_micro), start, m, out.data_ptr(),
                                            torch.cuda.current_stream(dev).cuda_stream))
        return out

    if not select:
        return make(first, count)
    kept = []
    for start in range(first, first + count, piece):
        part = make(start, min(piece, first + count - start))
        cx = ((part[:, 0].double() + 10.0) / 0.05).to(torch.int32)      # the reference's cell formula (main.c:134)
        kept.append(part[(cx >= x_lo) & (cx <= x_hi)])
        del part, cx
    return torch.cat(kept).contiguous() if kept else torch.empty((0, 3), dtype=torch.float32, device=dev)


def count_pairs(xyz, device="cuda:0", x_lo=0, x_hi=PC_NLAYERS):
    """Convenience: count pairs among host or device points on one GPU."""
    import torch
    if not isinstance(xyz, torch.Tensor):
        a, _, _ = _host_ptr(xyz)
        xyz = torch.from_numpy(a)
    xyz = xyz.to(device=device, dtype=torch.float32).contiguous().reshape(-1, 3)
    st = DeviceStages(xyz.shape[0], device)
    st.count_pairs(xyz, x_lo, x_hi)
    return st.pairs()


def parse_text_device(data, device="cuda:0", capacity=None):
    """Tokenizes text (bytes) on the GPU: returns (float32 tensor [n,3] on the device, records found, status bits)."""
    import torch
    raw = torch.frombuffer(bytearray(data) if len(data) else bytearray(1), dtype=torch.uint8)[:len(data)].to(device)
    if capacity is None:
        capacity = len(data) // 6 + 1
    out = torch.empty((capacity, 3), dtype=torch.float32, device=device)
    count = torch.zeros(1, dtype=torch.int64, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    wsb = lib().pc_parse_workspace_bytes(len(data))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=device)
    _check(lib().pc_parse_text_device(raw.data_ptr(), len(data), out.data_ptr(), capacity, count.data_ptr(),
                                      status.data_ptr(), ws.data_ptr(), wsb, torch.cuda.current_stream(out.device).cuda_stream))
    torch.cuda.synchronize()
    n = int(count.item())
    return out[:min(n, capacity)], n, int(status.item())
