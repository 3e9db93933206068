"""oracle -- the CHECKER, never the product.

ctypes bindings of oracle/liboracle.so (C restatement of the reference's filters, oracle.c) and,
when it has been built, oracle/_ref/libref_cpu_bilateral.so (the reference's own CPU loop,
src/main.cpp:1827-1864, compiled from /root/reference by oracle/Makefile).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Parity status: orc_cpu_bilateral is pinned bit-for-bit against _ref and tests/golden; the
shader restatements (a1-a6) are "parity unpinned" by the reference -- it has no tests, fixtures or a
runnable GPU build (see oracle.h).  What they are held against instead: reference-run CPU output on blue-constant
images (a1/a2/a3/a5/a6, tests/test_reference_fixtures.py) and known answers worked out by hand from the shaders' text
(NLM on step-edge frames, layer-guided bilateral with layers that differ from the image: tests/np_reference.py,
tests/test_oracle.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_cpu_bilateral.so")

_fp = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(quiet=True):
    """Compile liboracle.so (and _ref when /root/reference is present)."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _load():
    if not os.path.exists(_LIB):
        build()
    return ctypes.CDLL(_LIB)


_lib = _load()
_cf = ctypes.c_float


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def _hw(img):
    if img.ndim != 3 or img.shape[2] != 4:
        raise ValueError("image must be (h, w, 4)")
    return img.shape[0], img.shape[1]


def bilateral_texture(img, radius, sigma_s=2.0, sigma_c=0.2):
    img, p = _f32(img)
    h, w = _hw(img)
    out = np.empty_like(img)
    _lib.orc_bilateral_texture(p, w, h, radius, _cf(sigma_s), _cf(sigma_c), out.ctypes.data_as(_fp))
    return out


def bilateral_linear(img, radius, sigma_s=2.0, sigma_c=0.2):
    img, p = _f32(img)
    h, w = _hw(img)
    out = np.empty_like(img)
    _lib.orc_bilateral_linear(p, w, h, radius, _cf(sigma_s), _cf(sigma_c), out.ctypes.data_as(_fp))
    return out


def bilateral_layers_accum(img, layer_u8, W, radius, sigma_s=2.0, sigma_c=0.2):
    img, p = _f32(img)
    h, w = _hw(img)
    layer = np.ascontiguousarray(layer_u8, dtype=np.uint8)
    W = np.array(W, dtype=np.float32, order="C", copy=True)
    _lib.orc_bilateral_layers_accum(p, layer.ctypes.data_as(_u8p), w, h, radius, _cf(sigma_s), _cf(sigma_c),
                                    W.ctypes.data_as(ctypes.c_void_p))
    return W


def nlm_accum(target, neighbour, W, hparam=0.5, search=(-7, 7), patch=(-3, 3), threads=1):
    """threads > 1: the same invocations spread over OpenMP threads (identical results; bench.py cpu_baseline)."""
    target, pt = _f32(target)
    neighbour, pn = _f32(neighbour)
    h, w = _hw(target)
    W = np.array(W, dtype=np.float32, order="C", copy=True)
    if threads > 1:
        _lib.orc_nlm_accum_mt(pt, pn, w, h, _cf(hparam), search[0], search[1], patch[0], patch[1],
                              W.ctypes.data_as(ctypes.c_void_p), int(threads))
    else:
        _lib.orc_nlm_accum(pt, pn, w, h, _cf(hparam), search[0], search[1], patch[0], patch[1],
                           W.ctypes.data_as(ctypes.c_void_p))
    return W


def normalize(W):
    W = np.ascontiguousarray(W, dtype=np.float32)
    h, w = W.shape[:2]
    out = np.empty((h, w, 4), np.float32)
    _lib.orc_normalize(W.ctypes.data_as(ctypes.c_void_p), w, h, out.ctypes.data_as(_fp))
    return out


def unpack_u8(u8, flavour=0):
    u8 = np.ascontiguousarray(u8, dtype=np.uint8)
    out = np.empty(u8.shape, np.float32)
    fn = _lib.orc_unpack_u8_unorm if flavour == 0 else _lib.orc_unpack_u8_cpu
    fn(u8.ctypes.data_as(_u8p), ctypes.c_long(u8.size), out.ctypes.data_as(_fp))
    return out


def pack_u8(f32):
    f32, p = _f32(f32)
    out = np.empty(f32.shape, np.uint8)
    _lib.orc_pack_u8(p, ctypes.c_long(f32.size), out.ctypes.data_as(_u8p))
    return out


def cpu_bilateral(img, radius=10, sigma_s=10.0, sigma_c=0.2, blue_bug=True, threads=1):
    img, p = _f32(img)
    h, w = _hw(img)
    out = np.empty_like(img)
    _lib.orc_cpu_bilateral(p, w, h, radius, _cf(sigma_s), _cf(sigma_c), 1 if blue_bug else 0, threads,
                           out.ctypes.data_as(_fp))
    return out


def nlm_temporal(frames, k=0, hparam=0.5, search=(-7, 7), patch=(-3, 3), first=0, count=None, threads=1):
    """The multi-frame mode as the product defines it: for output t accumulate over frames
    max(0,t-k)..min(n-1,t+k) in ascending order with target = frame t, then normalize."""
    n = len(frames)
    count = n - first if count is None else count
    outs = []
    for t in range(first, first + count):
        h, w = frames[t].shape[:2]
        W = np.zeros((h, w, 8), np.float32)
        for f in range(max(0, t - k), min(n - 1, t + k) + 1):
            W = nlm_accum(frames[t], frames[f], W, hparam, search, patch, threads=threads)
        outs.append(normalize(W))
    return outs


def have_ref():
    return os.path.exists(_REF)


def have_ref_as_shipped():
    return os.path.exists(_REF.replace(".so", "_O0.so"))


def ref_cpu_bilateral(img, radius=10, threads=1, as_shipped=False):
    """The reference's own loop (sigma_s=10, sigma_c=0.2 are literals inside the slice).  as_shipped: the build with the
    reference's own compile flags ("-fopenmp -g", no optimisation level: CMakeLists.txt:31) instead of -O2."""
    path = _REF.replace(".so", "_O0.so") if as_shipped else _REF
    if not os.path.exists(path):
        raise RuntimeError(f"{path} has not been built (needs /root/reference)")
    ref = ctypes.CDLL(path)
    img, p = _f32(img)
    h, w = _hw(img)
    out = np.empty_like(img)
    ref.ref_cpu_bilateral(p, w, h, radius, threads, out.ctypes.data_as(_fp))
    return out
