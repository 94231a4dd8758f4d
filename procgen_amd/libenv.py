"""
ctypes binding of the gym3 *libenv* C ABI (include/libenv.h).

This plays the role of ``gym3.libenv.CEnv`` (the third-party cffi shim the reference
subclasses in reference procgen/env.py:66-136; gym3 is not installed in this image).
Call order is CEnv's: libenv_make -> libenv_get_tensortypes x3 -> allocate numpy
buffers -> libenv_set_buffers -> libenv_observe / libenv_act loop -> libenv_close.

Option encoding follows the reference's Python side (reference procgen/env.py:110-124,
237-245 via gym3): bool -> UINT8 count 1, int -> INT32 count 1, str -> UINT8 array of
len(str) bytes, NOT NUL-terminated (reference src/vecoptions.cpp:8-14).

The loader is library-agnostic: it drives the HIP ``libenv.so`` of this package and --
in the tests / oracle tooling -- the compiled reference ``oracle/_ref/libenv.so``.
"""
import ctypes as C
import os

import numpy as np

LIBENV_MAX_NAME_LEN = 128
LIBENV_MAX_NDIM = 16

DTYPE_UINT8, DTYPE_INT32, DTYPE_FLOAT32 = 1, 2, 3
SPACE_OBSERVATION, SPACE_ACTION, SPACE_INFO = 1, 2, 3

_NP_DTYPES = {DTYPE_UINT8: np.uint8, DTYPE_INT32: np.int32, DTYPE_FLOAT32: np.float32}


class _Value(C.Union):
    _fields_ = [("uint8", C.c_uint8), ("int32", C.c_int32), ("float32", C.c_float)]


class TensorType(C.Structure):
    _fields_ = [
        ("name", C.c_char * LIBENV_MAX_NAME_LEN),
        ("scalar_type", C.c_int),
        ("dtype", C.c_int),
        ("shape", C.c_int * LIBENV_MAX_NDIM),
        ("ndim", C.c_int),
        ("low", _Value),
        ("high", _Value),
    ]


class Option(C.Structure):
    _fields_ = [
        ("name", C.c_char * LIBENV_MAX_NAME_LEN),
        ("dtype", C.c_int),
        ("count", C.c_int),
        ("data", C.c_void_p),
    ]


class Options(C.Structure):
    _fields_ = [("items", C.POINTER(Option)), ("count", C.c_int)]


class Buffers(C.Structure):
    _fields_ = [
        ("ob", C.POINTER(C.c_void_p)),
        ("rew", C.POINTER(C.c_float)),
        ("first", C.POINTER(C.c_uint8)),
        ("info", C.POINTER(C.c_void_p)),
        ("ac", C.POINTER(C.c_void_p)),
    ]


def _bind(lib):
    lib.libenv_version.restype = C.c_int
    lib.libenv_version.argtypes = []
    lib.libenv_make.restype = C.c_void_p
    lib.libenv_make.argtypes = [C.c_int, Options]
    lib.libenv_get_tensortypes.restype = C.c_int
    lib.libenv_get_tensortypes.argtypes = [C.c_void_p, C.c_int, C.POINTER(TensorType)]
    lib.libenv_set_buffers.restype = None
    lib.libenv_set_buffers.argtypes = [C.c_void_p, C.POINTER(Buffers)]
    lib.libenv_observe.restype = None
    lib.libenv_observe.argtypes = [C.c_void_p]
    lib.libenv_act.restype = None
    lib.libenv_act.argtypes = [C.c_void_p]
    lib.libenv_close.restype = None
    lib.libenv_close.argtypes = [C.c_void_p]
    for name, restype in (("get_state", C.c_int), ("set_state", None)):
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]


class TensorSpec:
    """Plain description of one libenv_tensortype."""

    def __init__(self, tt):
        self.name = tt.name.decode()
        self.dtype = _NP_DTYPES[tt.dtype]
        self.shape = tuple(tt.shape[i] for i in range(tt.ndim))
        self.low = getattr(tt.low, {1: "uint8", 2: "int32", 3: "float32"}[tt.dtype])
        self.high = getattr(tt.high, {1: "uint8", 2: "int32", 3: "float32"}[tt.dtype])
        self.scalar_type = tt.scalar_type

    def __repr__(self):
        return f"TensorSpec({self.name!r}, {np.dtype(self.dtype).name}, {self.shape}, [{self.low},{self.high}])"


class CEnv:
    """Vectorized environment living in a libenv shared library."""

    def __init__(self, lib_dir, num, options, lib_name="libenv.so", buffer_padding=0):
        """buffer_padding > 0 leaves that many unused bytes behind every env's slice of the observation / action / info
        arrays: libenv only promises per-env pointers (libenv_buffers), so a library must not assume one dense array."""
        path = lib_dir if os.path.isfile(lib_dir) else os.path.join(lib_dir, lib_name)
        if not os.path.exists(path):
            raise FileNotFoundError(f"libenv library not found: {path}")
        self._lib = C.CDLL(path)
        _bind(self._lib)
        assert self._lib.libenv_version() == 1, "libenv version mismatch"
        self.num = int(num)
        self.options = dict(options)

        keep = []
        items = (Option * len(options))()
        for i, (k, v) in enumerate(options.items()):
            items[i].name = k.encode()
            if isinstance(v, (bool, np.bool_)):
                arr = np.array([int(v)], dtype=np.uint8)
                items[i].dtype = DTYPE_UINT8
            elif isinstance(v, (int, np.integer)):
                arr = np.array([v], dtype=np.int32)
                items[i].dtype = DTYPE_INT32
            elif isinstance(v, str):
                arr = np.frombuffer(v.encode(), dtype=np.uint8).copy()
                items[i].dtype = DTYPE_UINT8
            elif isinstance(v, float):
                arr = np.array([v], dtype=np.float32)
                items[i].dtype = DTYPE_FLOAT32
            else:
                raise TypeError(f"unsupported option type for {k}: {type(v)}")
            keep.append(arr)
            items[i].count = arr.size
            items[i].data = arr.ctypes.data
        opts = Options(items, len(options))
        self._handle = C.c_void_p(self._lib.libenv_make(self.num, opts))
        del keep
        if not self._handle:
            raise RuntimeError("libenv_make failed")

        self.ob_types = self._tensortypes(SPACE_OBSERVATION)
        self.ac_types = self._tensortypes(SPACE_ACTION)
        self.info_types = self._tensortypes(SPACE_INFO)

        n = self.num
        self._stores = []

        def alloc(t):
            if not buffer_padding:
                return np.zeros((n,) + t.shape, dtype=t.dtype)
            item = int(np.prod(t.shape, dtype=np.int64)) * np.dtype(t.dtype).itemsize
            pad = -(-(item + buffer_padding) // np.dtype(t.dtype).itemsize) * np.dtype(t.dtype).itemsize
            store = np.zeros((n, pad), dtype=np.uint8)
            self._stores.append(store)
            # a strided view: env e starts at e * pad bytes (built from the buffer directly: .view() on a non-contiguous
            # array needs numpy >= 1.23)
            inner = np.zeros(t.shape, dtype=t.dtype).strides
            return np.ndarray((n,) + tuple(t.shape), dtype=t.dtype, buffer=store, strides=(pad,) + tuple(inner))

        self._ob = {t.name: alloc(t) for t in self.ob_types}
        self._ac = {t.name: alloc(t) for t in self.ac_types}
        self._info = {t.name: alloc(t) for t in self.info_types}
        self._rew = np.zeros(n, dtype=np.float32)
        self._first = np.zeros(n, dtype=np.uint8)

        def table(types, arrays):
            tab = (C.c_void_p * (len(types) * n))()
            for s, t in enumerate(types):
                a = arrays[t.name]
                stride = a.strides[0] if n else 0
                for e in range(n):
                    tab[s * n + e] = a.ctypes.data + e * stride
            return tab

        self._ob_tab = table(self.ob_types, self._ob)
        self._ac_tab = table(self.ac_types, self._ac)
        self._info_tab = table(self.info_types, self._info)
        self._bufs = Buffers(
            C.cast(self._ob_tab, C.POINTER(C.c_void_p)),
            self._rew.ctypes.data_as(C.POINTER(C.c_float)),
            self._first.ctypes.data_as(C.POINTER(C.c_uint8)),
            C.cast(self._info_tab, C.POINTER(C.c_void_p)),
            C.cast(self._ac_tab, C.POINTER(C.c_void_p)),
        )
        self._lib.libenv_set_buffers(self._handle, C.byref(self._bufs))
        self._closed = False

    def _tensortypes(self, space):
        cnt = self._lib.libenv_get_tensortypes(self._handle, space, None)
        arr = (TensorType * max(cnt, 1))()
        self._lib.libenv_get_tensortypes(self._handle, space, arr)
        return [TensorSpec(arr[i]) for i in range(cnt)]

    # -- gym3-style interface ------------------------------------------------
    def observe(self):
        """Join the pending step; returns (rew, ob_dict, first) views over the registered buffers."""
        self._lib.libenv_observe(self._handle)
        return self._rew, self._ob, self._first.astype(bool)

    def act(self, ac):
        if isinstance(ac, dict):
            for k, v in ac.items():
                self._ac[k][...] = v
        else:
            self._ac[self.ac_types[0].name][...] = ac
        self._lib.libenv_act(self._handle)

    def get_info(self):
        self._lib.libenv_observe(self._handle)
        return [{k: v[e] for k, v in self._info.items()} for e in range(self.num)]

    def info_arrays(self):
        """Vectorized variant of get_info(): dict name -> array over envs (no per-env dicts)."""
        return self._info

    def call_c_func(self, name, *args):
        return getattr(self._lib, name)(self._handle, *args)

    def close(self):
        if not self._closed:
            self._lib.libenv_close(self._handle)
            self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
