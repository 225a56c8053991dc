"""ctypes binding of libroomnet_hip.so (C ABI in include/roomnet_hip.h).

There is deliberately no CPU fallback: if the HIP library is missing or cannot be
loaded, importing the product path fails loudly with ``RoomNetLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .graph import BN_EPSILON, Graph

RN_OK = 0
RN_DTYPE_F32, RN_DTYPE_BF16, RN_DTYPE_F16 = 0, 1, 2
RN_FLAG_TAPS = 1
RN_FLAG_STAGE_LAUNCHES = 2      # 16-bit handles: one launch per conv stage (no cross-stage fusion)
RN_FLAG_GENERIC_KERNELS = 4     # 16-bit handles: generic stage kernel everywhere (diagnostic cross-check)
RN_FLAG_PAIR_32X32 = 8          # 16-bit handles: the fused stage pair on the round-2 32x32x16 kernel (comparison arm)
RN_FLAG_COMPUTE_FROZEN = 16
RN_FLAG_NO_DITHER = 32     # convolve the provably constant channels too (comparison arm of the frozen-channel folding)
RN_MAX_STAGES = 16
RN_MAX_DENSE = 8
RN_NAME_LEN = 32

DTYPES = {"f32": RN_DTYPE_F32, "fp32": RN_DTYPE_F32, "float32": RN_DTYPE_F32,
          "bf16": RN_DTYPE_BF16, "bfloat16": RN_DTYPE_BF16,
          "f16": RN_DTYPE_F16, "fp16": RN_DTYPE_F16, "float16": RN_DTYPE_F16}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libroomnet_hip.so")
# the test / A-B library: the product library's objects + the round-2 comparison kernels behind RN_FLAG_PAIR_32X32
AB_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libroomnet_hip_ab.so")

# every symbol include/roomnet_hip.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = (
    "rn_create", "rn_destroy", "rn_last_error", "rn_device_count", "rn_version",
    "rn_forward_u8", "rn_submit_u8", "rn_collect", "rn_forward_f32", "rn_forward_u8_device", "rn_forward_f32_device", "rn_sync",
    "rn_set_stream", "rn_set_stream_null", "rn_node_count", "rn_node_info_get", "rn_tap", "rn_set_profiling", "rn_timing",
    "rn_dominant_stage", "rn_stage_launch", "rn_device_malloc", "rn_device_free", "rn_memcpy_h2d", "rn_memcpy_d2h",
    "rn_crop_resize_u8_device", "rn_crop_resize_batch_u8_device", "rn_classify_images_u8", "rn_host_alloc", "rn_host_free", "rn_frozen_info", "rn_const_info",
    "rn_group_create", "rn_group_destroy", "rn_group_size", "rn_group_handle", "rn_group_forward_u8",
    "rn_group_forward_u8_device", "rn_group_result_buffer", "rn_group_sync", "rn_group_plan",
)


class RoomNetLibraryError(RuntimeError):
    """libroomnet_hip.so is missing / unloadable, or a call into it failed."""


_fp = C.POINTER(C.c_float)


class rn_conv_stage(C.Structure):
    _fields_ = [("cin", C.c_int32), ("cout", C.c_int32), ("pool_k", C.c_int32), ("pool_s", C.c_int32),
                ("skip_stage", C.c_int32),
                ("kernel", _fp), ("gamma", _fp), ("beta", _fp), ("mean", _fp), ("variance", _fp),
                ("gamma2", _fp), ("beta2", _fp), ("mean2", _fp), ("variance2", _fp)]


class rn_dense_layer(C.Structure):
    _fields_ = [("nin", C.c_int32), ("nout", C.c_int32), ("kernel", _fp), ("bias", _fp),
                ("gamma", _fp), ("beta", _fp), ("mean", _fp), ("variance", _fp)]


class rn_weights(C.Structure):
    _fields_ = [("im_side", C.c_int32), ("num_classes", C.c_int32), ("n_stages", C.c_int32),
                ("n_dense", C.c_int32), ("bn_epsilon", C.c_float),
                ("stages", C.POINTER(rn_conv_stage)), ("dense", C.POINTER(rn_dense_layer))]


class rn_stage_ms(C.Structure):
    _fields_ = [("n_stages", C.c_int32), ("preprocess_ms", C.c_float),
                ("stage_ms", C.c_float * RN_MAX_STAGES), ("head_ms", C.c_float), ("total_ms", C.c_float)]


class rn_node_info(C.Structure):
    _fields_ = [("name", C.c_char * RN_NAME_LEN), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32)]


_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load libroomnet_hip.so and declare its prototypes.  Raises if unavailable."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("ROOMNET_HIP_LIB", LIB_PATH)
    if not os.path.isfile(p):
        raise RoomNetLibraryError(
            "libroomnet_hip.so not found at %r -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or roomnet_amd/csrc/build.sh" % p)
    try:
        lib = C.CDLL(p)
    except OSError as e:
        raise RoomNetLibraryError("cannot load %r: %s" % (p, e)) from e
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.rn_create.argtypes = [C.POINTER(rn_weights), i32, i32, i32, C.c_uint, C.POINTER(vp)]
    lib.rn_create.restype = i32
    lib.rn_destroy.argtypes = [vp]
    lib.rn_destroy.restype = None
    lib.rn_last_error.argtypes = []
    lib.rn_last_error.restype = C.c_char_p
    lib.rn_version.argtypes = []
    lib.rn_version.restype = C.c_char_p
    lib.rn_device_count.argtypes = []
    lib.rn_device_count.restype = i32
    for name in ("rn_forward_u8", "rn_forward_f32", "rn_forward_u8_device", "rn_forward_f32_device"):
        fn = getattr(lib, name)
        fn.argtypes = [vp, vp, i32, vp, vp]
        fn.restype = i32
    lib.rn_submit_u8.argtypes = [vp, vp, i32, i32]
    lib.rn_submit_u8.restype = i32
    lib.rn_collect.argtypes = [vp, i32, vp, vp]
    lib.rn_collect.restype = i32
    lib.rn_sync.argtypes = [vp]
    lib.rn_sync.restype = i32
    lib.rn_set_stream.argtypes = [vp, vp]
    lib.rn_set_stream.restype = i32
    lib.rn_set_stream_null.argtypes = [vp]
    lib.rn_set_stream_null.restype = i32
    lib.rn_node_count.argtypes = [vp]
    lib.rn_node_count.restype = i32
    lib.rn_node_info_get.argtypes = [vp, i32, C.POINTER(rn_node_info)]
    lib.rn_node_info_get.restype = i32
    lib.rn_tap.argtypes = [vp, i32, vp, sz, C.POINTER(sz)]
    lib.rn_tap.restype = i32
    lib.rn_set_profiling.argtypes = [vp, i32]
    lib.rn_set_profiling.restype = i32
    lib.rn_timing.argtypes = [vp, C.POINTER(rn_stage_ms)]
    lib.rn_timing.restype = i32
    lib.rn_dominant_stage.argtypes = [vp]
    lib.rn_dominant_stage.restype = i32
    lib.rn_stage_launch.argtypes = [vp, i32]
    lib.rn_stage_launch.restype = i32
    lib.rn_device_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.rn_device_malloc.restype = i32
    lib.rn_device_free.argtypes = [vp, vp]
    lib.rn_device_free.restype = i32
    lib.rn_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    lib.rn_memcpy_h2d.restype = i32
    if hasattr(lib, "rn_const_info"):
        lib.rn_const_info.argtypes = [vp, C.POINTER(C.c_int)]
        lib.rn_const_info.restype = i32
    if hasattr(lib, "rn_frozen_info"):
        lib.rn_frozen_info.argtypes = [vp, C.POINTER(C.c_int)]
        lib.rn_frozen_info.restype = i32
    if hasattr(lib, "rn_host_alloc"):           # (absent from the older libraries tools/gpu_var.sh loads through ROOMNET_HIP_LIB as A/B arms)
        lib.rn_host_alloc.argtypes = [sz, C.POINTER(vp)]
        lib.rn_host_alloc.restype = i32
        lib.rn_host_free.argtypes = [vp]
        lib.rn_host_free.restype = i32
    lib.rn_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    lib.rn_memcpy_d2h.restype = i32
    lib.rn_crop_resize_u8_device.argtypes = [vp, vp, i32, i32, vp, i32]
    lib.rn_crop_resize_u8_device.restype = i32
    if hasattr(lib, "rn_crop_resize_batch_u8_device"):
        lib.rn_crop_resize_batch_u8_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int), i32, vp]
        lib.rn_crop_resize_batch_u8_device.restype = i32
    lib.rn_classify_images_u8.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int), i32, vp, vp]
    lib.rn_classify_images_u8.restype = i32
    lib.rn_group_create.argtypes = [C.POINTER(rn_weights), i32, C.POINTER(C.c_int), i32, i32, C.c_uint, C.POINTER(vp)]
    lib.rn_group_create.restype = i32
    lib.rn_group_destroy.argtypes = [vp]
    lib.rn_group_destroy.restype = None
    lib.rn_group_size.argtypes = [vp]
    lib.rn_group_size.restype = i32
    lib.rn_group_handle.argtypes = [vp, i32]
    lib.rn_group_handle.restype = vp
    lib.rn_group_forward_u8.argtypes = [vp, vp, i32, vp, vp]
    lib.rn_group_forward_u8.restype = i32
    lib.rn_group_forward_u8_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int)]
    lib.rn_group_forward_u8_device.restype = i32
    lib.rn_group_result_buffer.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz)]
    lib.rn_group_result_buffer.restype = i32
    lib.rn_group_sync.argtypes = [vp]
    lib.rn_group_sync.restype = i32
    if hasattr(lib, "rn_group_plan"):
        lib.rn_group_plan.argtypes = [i32, i32, i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(sz)]
        lib.rn_group_plan.restype = i32
    if path is None:
        _lib = lib
    return lib


def _check(lib: C.CDLL, rc: int, what: str) -> None:
    if rc != RN_OK:
        msg = lib.rn_last_error()
        text = msg.decode("utf-8", "replace") if msg else ""
        if rc in (-1, -5):
            raise ValueError("%s failed (%d): %s" % (what, rc, text))
        raise RoomNetLibraryError("%s failed (%d): %s" % (what, rc, text))


def device_count() -> int:
    return int(load_library().rn_device_count())


class _Packed:
    """Keeps the float32 arrays alive while the C structs point into them."""

    def __init__(self, graph: Graph, weights: Dict[str, np.ndarray]):
        self.keep: List[np.ndarray] = []
        self.stages = (rn_conv_stage * len(graph.stages))()
        self.dense = (rn_dense_layer * len(graph.dense))()

        def ptr(name: str, shape: Tuple[int, ...]):
            if name not in weights:
                raise KeyError("tensor %r not found in checkpoint" % name)
            a = np.ascontiguousarray(weights[name], dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise ValueError("tensor %r has shape %s, the graph needs %s"
                                 % (name, tuple(a.shape), tuple(shape)))
            self.keep.append(a)
            return a.ctypes.data_as(_fp)

        for i, s in enumerate(graph.stages):
            st = self.stages[i]
            st.cin, st.cout, st.pool_k, st.pool_s, st.skip_stage = s.cin, s.cout, s.pool_k, s.pool_s, s.skip_stage
            st.kernel = ptr(s.conv_name + "/kernel", (3, 3, s.cin, s.cout))
            st.gamma = ptr(s.bn_name + "/gamma", (s.cout,))
            st.beta = ptr(s.bn_name + "/beta", (s.cout,))
            st.mean = ptr(s.bn_name + "/moving_mean", (s.cout,))
            st.variance = ptr(s.bn_name + "/moving_variance", (s.cout,))
            if s.residual:
                st.gamma2 = ptr(s.bn2_name + "/gamma", (s.cout,))
                st.beta2 = ptr(s.bn2_name + "/beta", (s.cout,))
                st.mean2 = ptr(s.bn2_name + "/moving_mean", (s.cout,))
                st.variance2 = ptr(s.bn2_name + "/moving_variance", (s.cout,))
        for i, d in enumerate(graph.dense):
            dl = self.dense[i]
            dl.nin, dl.nout = d.nin, d.nout
            dl.kernel = ptr(d.name + "/kernel", (d.nin, d.nout))
            if d.biased:
                dl.bias = ptr(d.name + "/bias", (d.nout,))
            if d.bn_name:
                dl.gamma = ptr(d.bn_name + "/gamma", (d.nout,))
                dl.beta = ptr(d.bn_name + "/beta", (d.nout,))
                dl.mean = ptr(d.bn_name + "/moving_mean", (d.nout,))
                dl.variance = ptr(d.bn_name + "/moving_variance", (d.nout,))
        self.w = rn_weights()
        self.w.im_side = graph.im_side
        self.w.num_classes = graph.num_classes
        self.w.n_stages = len(graph.stages)
        self.w.n_dense = len(graph.dense)
        self.w.bn_epsilon = BN_EPSILON
        self.w.stages = C.cast(self.stages, C.POINTER(rn_conv_stage))
        self.w.dense = C.cast(self.dense, C.POINTER(rn_dense_layer))


class Engine:
    """One rn_handle: a model instance bound to one GPU and one stream."""

    def __init__(self, graph: Graph, weights: Dict[str, np.ndarray], device: int = 0, dtype="f32",
                 max_batch: int = 64, taps: bool = False, lib_path: Optional[str] = None,
                 stage_launches: bool = False, generic_kernels: bool = False, pair32: bool = False,
                 compute_frozen: bool = False, no_dither: bool = False):
        if pair32 and lib_path is None and "ROOMNET_HIP_LIB" not in os.environ:
            lib_path = AB_LIB_PATH           # (the round-2 comparison kernels are not in the product library)
        self.lib = load_library(lib_path)
        self.graph = graph
        self.dtype = DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
        self.max_batch = int(max_batch)
        self.device = int(device)
        packed = _Packed(graph, weights)
        h = C.c_void_p()
        rc = self.lib.rn_create(C.byref(packed.w), self.device, self.dtype, self.max_batch,
                                (RN_FLAG_TAPS if taps else 0) | (RN_FLAG_STAGE_LAUNCHES if stage_launches else 0)
                                | (RN_FLAG_GENERIC_KERNELS if generic_kernels else 0)
                                | (RN_FLAG_PAIR_32X32 if pair32 else 0)
                                | (RN_FLAG_COMPUTE_FROZEN if compute_frozen else 0)
                                | (RN_FLAG_NO_DITHER if no_dither else 0), C.byref(h))
        _check(self.lib, rc, "rn_create")
        self._h = h
        self._nodes: Optional[Dict[str, Tuple[int, Tuple[int, int, int]]]] = None

    # -- lifetime
    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.rn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self) -> C.c_void_p:
        if not self._h:
            raise RoomNetLibraryError("engine is closed")
        return self._h

    # -- forward
    def forward_u8(self, im_bgr_u8: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        s = self.graph.im_side
        im = np.ascontiguousarray(im_bgr_u8, dtype=np.uint8)
        if im.ndim != 4 or im.shape[1:] != (s, s, 3):
            raise ValueError("expected a [N,%d,%d,3] uint8 batch, got %s" % (s, s, im.shape))
        n = im.shape[0]
        probs = np.empty((n, self.graph.num_classes), np.float32)
        ids = np.empty((n,), np.int64)
        for i in range(0, n, self.max_batch):
            m = min(self.max_batch, n - i)
            rc = self.lib.rn_forward_u8(self.handle, im[i:i + m].ctypes.data, m, probs[i:i + m].ctypes.data,
                                        ids[i:i + m].ctypes.data)
            _check(self.lib, rc, "rn_forward_u8")
        return ids, probs

    def classify_images(self, images) -> Tuple[np.ndarray, np.ndarray]:
        """``images``: sequence of BGR uint8 HWC arrays of any (individual) size.  Centre crop + INTER_LINEAR resize
        run on the GPU (``rn_classify_images_u8``); returns ``(ids [n], probs [n, C])``."""
        ims = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        for im in ims:
            if im.ndim != 3 or im.shape[2] != 3 or im.shape[0] < 1 or im.shape[1] < 1:
                raise ValueError("expected HWC uint8 images with 3 channels, got %s" % (im.shape,))
        n = len(ims)
        probs = np.empty((n, self.graph.num_classes), np.float32)
        ids = np.empty((n,), np.int64)
        for i in range(0, n, self.max_batch):
            chunk = ims[i:i + self.max_batch]
            m = len(chunk)
            ptrs = (C.c_void_p * m)(*[im.ctypes.data for im in chunk])
            hs = (C.c_int * m)(*[im.shape[0] for im in chunk])
            ws = (C.c_int * m)(*[im.shape[1] for im in chunk])
            rc = self.lib.rn_classify_images_u8(self.handle, ptrs, hs, ws, m, probs[i:i + m].ctypes.data,
                                                ids[i:i + m].ctypes.data)
            _check(self.lib, rc, "rn_classify_images_u8")
        return ids, probs

    def crop_resize(self, im_bgr_u8: np.ndarray) -> np.ndarray:
        """One image through the device crop + resize; returns the ``[S, S, 3]`` uint8 result (parity tests)."""
        im = np.ascontiguousarray(im_bgr_u8, dtype=np.uint8)
        s = self.graph.im_side
        d_src = self.device_malloc(im.nbytes)
        d_dst = self.device_malloc(self.max_batch * s * s * 3)
        try:
            self.h2d(d_src, im)
            rc = self.lib.rn_crop_resize_u8_device(self.handle, C.c_void_p(d_src), im.shape[0], im.shape[1],
                                                   C.c_void_p(d_dst), 0)
            _check(self.lib, rc, "rn_crop_resize_u8_device")
            out = np.empty((s, s, 3), np.uint8)
            self.d2h(out, d_dst)
            return out
        finally:
            self.device_free(d_src)
            self.device_free(d_dst)

    def crop_resize_batch(self, ims, repeat: int = 1) -> np.ndarray:
        """A list of images (any sizes, <= max_batch of them) through the BATCHED device crop + resize -- one launch for all of them
        (``rn_crop_resize_batch_u8_device``); returns the ``[n, S, S, 3]`` uint8 results.  ``repeat`` > 1 launches it that often
        (profiling / timing of the one kernel)."""
        ims = [np.ascontiguousarray(im, dtype=np.uint8) for im in ims]
        n, s = len(ims), self.graph.im_side
        d_srcs = [self.device_malloc(im.nbytes) for im in ims]
        d_dst = self.device_malloc(self.max_batch * s * s * 3)
        try:
            for d, im in zip(d_srcs, ims):
                self.h2d(d, im)
            ptrs = (C.c_void_p * n)(*d_srcs)
            hs = (C.c_int * n)(*[im.shape[0] for im in ims])
            ws = (C.c_int * n)(*[im.shape[1] for im in ims])
            for _ in range(max(1, repeat)):
                _check(self.lib, self.lib.rn_crop_resize_batch_u8_device(self.handle, ptrs, hs, ws, n, C.c_void_p(d_dst)),
                       "rn_crop_resize_batch_u8_device")
            self.sync()
            out = np.empty((n, s, s, 3), np.uint8)
            self.d2h(out, d_dst)
            return out
        finally:
            for d in d_srcs:
                self.device_free(d)
            self.device_free(d_dst)

    def forward_f32(self, x_rgb: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        s = self.graph.im_side
        x = np.ascontiguousarray(x_rgb, dtype=np.float32)
        if x.ndim != 4 or x.shape[1:] != (s, s, 3):
            raise ValueError("expected a [N,%d,%d,3] float32 batch, got %s" % (s, s, x.shape))
        n = x.shape[0]
        probs = np.empty((n, self.graph.num_classes), np.float32)
        ids = np.empty((n,), np.int64)
        for i in range(0, n, self.max_batch):
            m = min(self.max_batch, n - i)
            rc = self.lib.rn_forward_f32(self.handle, x[i:i + m].ctypes.data, m, probs[i:i + m].ctypes.data,
                                         ids[i:i + m].ctypes.data)
            _check(self.lib, rc, "rn_forward_f32")
        return ids, probs

    def forward_u8_device(self, d_bgr: int, n: int, d_probs: int, d_ids: int) -> None:
        """Asynchronous: raw device pointers (e.g. ``tensor.data_ptr()``)."""
        rc = self.lib.rn_forward_u8_device(self.handle, C.c_void_p(d_bgr), n, C.c_void_p(d_probs),
                                           C.c_void_p(d_ids))
        _check(self.lib, rc, "rn_forward_u8_device")

    def sync(self) -> None:
        _check(self.lib, self.lib.rn_sync(self.handle), "rn_sync")

    def submit_u8(self, bgr_nhwc: np.ndarray, slot: int) -> None:
        """Upload + enqueue one batch (<= max_batch images) into pipeline slot 0 or 1 (``rn_submit_u8``)."""
        s = self.graph.im_side
        x = np.ascontiguousarray(bgr_nhwc, dtype=np.uint8)
        if x.ndim != 4 or x.shape[1:] != (s, s, 3):
            raise ValueError("expected uint8 [n,%d,%d,3], got %s" % (s, s, x.shape))
        _check(self.lib, self.lib.rn_submit_u8(self.handle, x.ctypes.data, x.shape[0], slot), "rn_submit_u8")
        self._slot_n = getattr(self, "_slot_n", {})
        self._slot_n[slot] = x.shape[0]

    def collect(self, slot: int) -> Tuple[np.ndarray, np.ndarray]:
        """Wait for the batch in `slot` and return ``(ids, probs)`` (``rn_collect``)."""
        n = getattr(self, "_slot_n", {}).get(slot, 0)
        probs = np.empty((max(n, 1), self.graph.num_classes), np.float32)
        ids = np.empty((max(n, 1),), np.int64)
        _check(self.lib, self.lib.rn_collect(self.handle, slot, probs.ctypes.data, ids.ctypes.data), "rn_collect")
        return ids[:n], probs[:n]

    def set_stream(self, hip_stream: Optional[int]) -> None:
        """Run on the given hipStream_t handle; ``None`` restores the engine's own (non-blocking) stream; ``0`` selects
        the HIP null stream (what ``torch.cuda.current_stream().cuda_stream`` is when no stream context is active)."""
        if hip_stream is None:
            _check(self.lib, self.lib.rn_set_stream(self.handle, C.c_void_p(0)), "rn_set_stream")
        elif hip_stream == 0:
            _check(self.lib, self.lib.rn_set_stream_null(self.handle), "rn_set_stream_null")
        else:
            _check(self.lib, self.lib.rn_set_stream(self.handle, C.c_void_p(hip_stream)), "rn_set_stream")

    # -- device memory helpers
    def device_malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _check(self.lib, self.lib.rn_device_malloc(self.handle, nbytes, C.byref(p)), "rn_device_malloc")
        return int(p.value)

    def device_free(self, ptr: int) -> None:
        _check(self.lib, self.lib.rn_device_free(self.handle, C.c_void_p(ptr)), "rn_device_free")

    def h2d(self, d_dst: int, src: np.ndarray) -> None:
        a = np.ascontiguousarray(src)
        _check(self.lib, self.lib.rn_memcpy_h2d(self.handle, C.c_void_p(d_dst), a.ctypes.data, a.nbytes),
               "rn_memcpy_h2d")

    def d2h(self, dst: np.ndarray, d_src: int) -> None:
        assert dst.flags["C_CONTIGUOUS"]
        _check(self.lib, self.lib.rn_memcpy_d2h(self.handle, dst.ctypes.data, C.c_void_p(d_src), dst.nbytes),
               "rn_memcpy_d2h")

    # -- introspection
    def nodes(self) -> Dict[str, Tuple[int, Tuple[int, int, int]]]:
        if self._nodes is None:
            out = {}
            for i in range(self.lib.rn_node_count(self.handle)):
                info = rn_node_info()
                _check(self.lib, self.lib.rn_node_info_get(self.handle, i, C.byref(info)), "rn_node_info_get")
                out[info.name.decode()] = (i, (info.h, info.w, info.c))
            self._nodes = out
        return self._nodes

    def tap(self, name: str, n: int) -> np.ndarray:
        nid, (h, w, c) = self.nodes()[name]
        out = np.empty((n, h, w, c), np.float32)
        got = C.c_size_t()
        _check(self.lib, self.lib.rn_tap(self.handle, nid, out.ctypes.data, out.size, C.byref(got)), "rn_tap")
        if got.value != out.size:
            raise RoomNetLibraryError("rn_tap(%s): expected %d elements, library has %d" % (name, out.size, got.value))
        if h == 1 and w == 1:
            return out.reshape(n, c)
        return out

    def frozen_info(self) -> Dict[str, int]:
        """What rn_create folded on this handle (``rn_frozen_info``): channels of the first 32 -> 32 stage's output (16-bit
        handles: the fused pair's on-chip tensor) that are provably constant and not convolved / not contracted by the next
        stage, how many were proven, the residual stage whose frozen first-BN channels are folded and how many of its 16-cout
        quarters still run their convolution."""
        info = (C.c_int * 4)(0, 0, -1, 4)
        if hasattr(self.lib, "rn_frozen_info"):      # (older libraries loaded as A/B arms fold nothing)
            _check(self.lib, self.lib.rn_frozen_info(self.handle, info), "rn_frozen_info")
        return {"pair_channels_not_convolved": info[0], "pair_channels_proven_frozen": info[1], "residual_stage_folded": info[2],
                "residual_stage_live_quarters": info[3]}

    def const_info(self) -> Dict[str, int]:
        """Constant channels nobody computes on this handle (``rn_const_info``): the conv stage whose last 16 output channels
        are constants in the handle's 16-bit store (written once at rn_create), how many of its channels were proven so, how
        many are folded, and how many input channels the stage behind it still contracts."""
        info = (C.c_int * 4)(-1, 0, 0, 0)
        if hasattr(self.lib, "rn_const_info"):       # (older libraries loaded as A/B arms fold nothing)
            _check(self.lib, self.lib.rn_const_info(self.handle, info), "rn_const_info")
        return {"stage": info[0], "channels_proven_constant": info[1], "channels_not_convolved": info[2],
                "next_stage_input_channels": info[3]}

    def set_profiling(self, enable: bool) -> None:
        _check(self.lib, self.lib.rn_set_profiling(self.handle, 1 if enable else 0), "rn_set_profiling")

    def timing(self) -> Dict[str, object]:
        t = rn_stage_ms()
        _check(self.lib, self.lib.rn_timing(self.handle, C.byref(t)), "rn_timing")
        return {"preprocess_ms": t.preprocess_ms, "stage_ms": [t.stage_ms[i] for i in range(t.n_stages)],
                "head_ms": t.head_ms, "total_ms": t.total_ms}

    def dominant_stage(self) -> int:
        return int(self.lib.rn_dominant_stage(self.handle))

    def launch_groups(self):
        """Conv stages grouped by launch: [[0], [1], [2, 3], [4], ...] when stages 2 and 3 run as one kernel.  A group's
        time is reported under its last stage in `timing()["stage_ms"]`."""
        groups = {}
        for i in range(len(self.graph.stages)):
            rep = int(self.lib.rn_stage_launch(self.handle, i))
            if rep < 0:
                raise RoomNetLibraryError(self.lib.rn_last_error().decode())
            groups.setdefault(rep, []).append(i)
        return [groups[k] for k in sorted(groups)]


class PinnedArray:
    """A NumPy array over page-locked host memory (``rn_host_alloc``): uploads out of it are asynchronous DMAs, so the
    two-slot pipeline (``Engine.submit_u8`` / ``collect``) hides them behind the previous batch's kernels.  Fill
    ``.array`` in place; ``close()`` (or garbage collection) frees the memory -- do not use ``.array`` afterwards."""

    def __init__(self, shape, dtype=np.uint8, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _check(self.lib, self.lib.rn_host_alloc(max(nbytes, 1), C.byref(p)), "rn_host_alloc")
        self._p = p
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self) -> None:
        if self._p:
            self.array = None
            self.lib.rn_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def group_plan(n: int, ndev: int, max_batch_per_device: int, num_classes: int = 6, lib_path: Optional[str] = None):
    """``rn_group_plan``: (counts, offsets, slot_bytes) of an n-image call on ndev devices -- no GPU needed."""
    lib = load_library(lib_path)
    counts, offsets, slot = (C.c_int * ndev)(), (C.c_int * ndev)(), C.c_size_t(0)
    _check(lib, lib.rn_group_plan(n, ndev, max_batch_per_device, num_classes, counts, offsets, C.byref(slot)), "rn_group_plan")
    return list(counts), list(offsets), int(slot.value)


class Group:
    """One rn_group: the model replicated on several GPUs of this process, batches sharded contiguously, one RCCL
    all-gather of the packed results (the C ABI's multi-GPU entry, include/roomnet_hip.h)."""

    def __init__(self, graph: Graph, weights: Dict[str, np.ndarray], devices: Sequence[int], dtype="bf16",
                 max_batch_per_device: int = 64, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        self.graph = graph
        self.devices = [int(d) for d in devices]
        self.cap = int(max_batch_per_device)
        packed = _Packed(graph, weights)
        dt = DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
        devs = (C.c_int * len(self.devices))(*self.devices)
        g = C.c_void_p()
        rc = self.lib.rn_group_create(C.byref(packed.w), len(self.devices), devs, dt, self.cap, 0, C.byref(g))
        _check(self.lib, rc, "rn_group_create")
        self._g = g

    def forward_u8(self, bgr_nhwc: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        s = self.graph.im_side
        x = np.ascontiguousarray(bgr_nhwc, dtype=np.uint8)
        if x.ndim != 4 or x.shape[1:] != (s, s, 3):
            raise ValueError("expected uint8 [n,%d,%d,3], got %s" % (s, s, x.shape))
        n = x.shape[0]
        probs = np.empty((n, self.graph.num_classes), np.float32)
        ids = np.empty((n,), np.int64)
        step = self.cap * len(self.devices)
        for i in range(0, n, step):
            m = min(step, n - i)
            rc = self.lib.rn_group_forward_u8(self._g, x[i:i + m].ctypes.data, m, probs[i:i + m].ctypes.data,
                                              ids[i:i + m].ctypes.data)
            _check(self.lib, rc, "rn_group_forward_u8")
        return ids, probs

    def close(self) -> None:
        if self._g:
            self.lib.rn_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
