"""ctypes binding of ``libmmwgpu.so`` (C ABI in ``include/mmwgpu.h``).

This is the ONLY compute backend of the package: there is no NumPy fallback.
If the shared library is missing, or no gfx950 device is visible, every
processor / detector raises ``MmwGpuError`` -- loudly, by design.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMWGPU_LIB: load another build of the same library (A/B timing of kernel variants; never a different backend)
LIB_PATH = os.environ.get("MMWGPU_LIB") or os.path.join(_HERE, "csrc", "libmmwgpu.so")

MMW_OK = 0
MMW_ERR_INVALID = -1
MMW_ERR_TRUNCATED = -4
MMW_ERR_UNSUPPORTED = -5
RESULT_POOL_CAP = 1 << 30       # bytes of pinned result blocks the default context may hold (handed out + pooled)
ABI_VERSION = 5          # include/mmwgpu.h MMWGPU_ABI_VERSION: the argtypes below are for exactly this revision
CFAR_CA, CFAR_OS, CFAR_GO, CFAR_SO = 0, 1, 2, 3
ANGLE_MAGNITUDE, ANGLE_NO_WINDOW, ANGLE_NO_SHIFT = 1, 2, 4
QUEUE_COMPUTE, QUEUE_COPY = 0, 1


class MmwGpuError(RuntimeError):
    """Raised for any non-zero status of the C ABI, or when the HIP library cannot be used."""


_lib = None
_lib_lock = threading.Lock()

_vp, _i, _d, _f, _sz = C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_size_t
_ip = C.POINTER(C.c_int)

# name -> argtypes; every function returns int except the two string getters
_SIGNATURES = {
    "mmw_device_count": [_ip],
    "mmw_device_info": [_i, C.c_char_p, _i, C.c_char_p, _i, _ip, C.POINTER(_sz)],
    "mmw_ctx_create": [C.POINTER(_vp), _i],
    "mmw_ctx_destroy": [_vp],
    "mmw_sync": [_vp],
    "mmw_malloc": [_vp, C.POINTER(_vp), _sz],
    "mmw_free": [_vp, _vp],
    "mmw_memcpy_h2d": [_vp, _vp, _vp, _sz],
    "mmw_memcpy_d2h": [_vp, _vp, _vp, _sz],
    "mmw_memset": [_vp, _vp, _i, _sz],
    "mmw_diag_set_option": [_vp, C.c_char_p, _i],
    "mmw_host_alloc": [_vp, C.POINTER(_vp), _sz],
    "mmw_host_free": [_vp, _vp],
    "mmw_memcpy_async": [_vp, _vp, _vp, _sz, _i, _i],
    "mmw_event_create": [_vp, C.POINTER(_vp)],
    "mmw_event_destroy": [_vp, _vp],
    "mmw_event_record": [_vp, _vp, _i],
    "mmw_queue_wait_event": [_vp, _i, _vp],
    "mmw_event_sync": [_vp, _vp],
    "mmw_timer_start": [_vp],
    "mmw_timer_stop": [_vp, C.POINTER(_f)],
    "mmw_synth_cubes": [_vp, _vp, _i, _i, _i, _i, C.c_uint64, _i, _f],
    "mmw_virtual_array_reformat": [_vp, _vp, _vp, _i, _i, _i, _i, _i],
    "mmw_virtual_array_reformat_i16": [_vp, _vp, _vp, _i, _i, _i, _i, _i],
    "mmw_range_doppler": [_vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "mmw_range_doppler_mag64": [_vp, _vp, _vp, _i, _i, _i, _i, _i],
    "mmw_angle_fft": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "mmw_chain3d": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "mmw_range_doppler_raw": [_vp, _vp, _vp, _i, _i, _i, _i, _i],
    "mmw_chain3d_raw": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i],
    "mmw_range_doppler_raw_i16": [_vp, _vp, _vp, _i, _i, _i, _i, _i],
    "mmw_chain3d_raw_i16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i],
    "mmw_dbs_gather": [_vp, _vp, _ip, _ip, _vp, _i, _i, _i, _i, _i],
    "mmw_mean_over_range": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "mmw_doppler_azimuth": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i],
    "mmw_doppler_azimuth_zoom": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_d), _i, _i],
    "mmw_range_profile": [_vp, _vp, _vp, _i, _i, _i, _i, _i],
    "mmw_range_profile_f64": [_vp, _vp, _vp, _i, _i, _i, _i, _i],
    "mmw_range_zoom": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _d, _d],
    "mmw_range_angle": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _ip, _i, _i],
    "mmw_cfar2d": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _d, _i],
    "mmw_cfar1d": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _i],
    "mmw_compact2d": [_vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "mmw_detect_batch": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _d, _i, _i],
    "mmw_detect_points_supported": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i],
    "mmw_detect_points": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _d, _i, _i,
                          _ip, _i, _i, _ip, _i, _i, _i, _ip],
    "mmw_angle_argmax": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ip, _i, _i, _i],
    "mmw_plane_l1": [_vp, _vp, _vp, _i, _i, _i, _i],
    "mmw_angle_argmax_exact": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ip, _i, _i, _i, _ip],
    "mmw_angle_argmax_cells64": [_vp, _vp, _vp, _i, _i, _i, _i],
    "mmw_bartlett": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _d],
    "mmw_capon": [_vp, _vp, C.POINTER(_d), _vp, _i, _i, _i, _i, _i, _d],
    "mmw_abs_c64": [_vp, _vp, _vp, _sz],
    "mmw_widen_f32_f64": [_vp, _vp, _vp, _sz],
    "mmw_diag_membw": [_vp, _vp, _vp, _sz, _i, _i],
    "mmw_diag_mfma_peak": [_vp, _i, C.POINTER(_d)],
    "mmw_diag_rd_plan": [_i, _i, _i, _ip],
    "mmw_diag_chain_plan": [_vp, _i, _i, _i, _i, _i, _i, _ip],
    "mmw_diag_chain_plan_nodev": [_i, _i, _i, _i, _i, _i, _i, _i, _ip],
    "mmw_diag_detect_plan": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _ip],
    "mmw_diag_czt_runs": [C.POINTER(_d), _i, _i, _ip, _i, _ip],
    "mmw_profile_enable": [_vp, _i],
    "mmw_profile_get": [_vp, C.c_char_p, C.POINTER(_f), _ip],
    "mmw_profile_reset": [_vp],
}
EXPORTED = tuple(sorted(list(_SIGNATURES) + ["mmw_version", "mmw_last_error", "mmw_abi_version"]))


def load_library():
    """dlopen the in-tree HIP library and declare prototypes; raises if it is not built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MmwGpuError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C mmwave_radar_processing_amd/csrc` (needs hipcc, --offload-arch=gfx950)")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the host
            raise MmwGpuError(f"cannot load {LIB_PATH}: {e}") from e
        try:
            lib.mmw_abi_version.restype = C.c_int
            abi = lib.mmw_abi_version()
        except AttributeError:
            abi = None
        if abi != ABI_VERSION:
            raise MmwGpuError(f"{LIB_PATH} has ABI revision {abi}, this binding is written for {ABI_VERSION}: rebuild the "
                              "library (make -C mmwave_radar_processing_amd/csrc) or unset MMWGPU_LIB")
        for name, args in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        lib.mmw_version.restype = C.c_char_p
        lib.mmw_last_error.restype = C.c_char_p
        _lib = lib
        return lib


def check(status: int, ok_truncated: bool = False) -> int:
    if status == MMW_OK or (ok_truncated and status == MMW_ERR_TRUNCATED):
        return status
    msg = load_library().mmw_last_error().decode("utf-8", "replace")
    if status == -1:
        raise ValueError(msg)       # the reference raises ValueError for bad arguments
    raise MmwGpuError(f"libmmwgpu status {status}: {msg}")


def device_count() -> int:
    n = C.c_int(0)
    check(load_library().mmw_device_count(C.byref(n)))
    return n.value


def device_info(device: int = 0) -> dict:
    name = C.create_string_buffer(256)
    arch = C.create_string_buffer(256)
    ncu = C.c_int(0)
    mem = C.c_size_t(0)
    check(load_library().mmw_device_info(device, name, 256, arch, 256, C.byref(ncu), C.byref(mem)))
    return dict(name=name.value.decode(), arch=arch.value.decode(), num_cu=ncu.value, total_mem=mem.value)


class DeviceBuffer:
    """A block of HBM owned by a Context; ``ptr`` is the raw device address."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(ctx.lib.mmw_malloc(ctx.handle, C.byref(p), self.nbytes))
        self.ptr = p.value

    def at(self, byte_offset: int) -> int:
        return self.ptr + int(byte_offset)

    def upload(self, arr: np.ndarray, byte_offset: int = 0):
        arr = np.ascontiguousarray(arr)
        if byte_offset + arr.nbytes > self.nbytes:
            raise ValueError("upload exceeds device buffer")
        check(self.ctx.lib.mmw_memcpy_h2d(self.ctx.handle, self.ptr + byte_offset, arr.ctypes.data, arr.nbytes))

    def download(self, shape, dtype, byte_offset: int = 0) -> np.ndarray:
        out = self.ctx.result_array(shape, dtype)
        if byte_offset + out.nbytes > self.nbytes:
            raise ValueError("download exceeds device buffer")
        check(self.ctx.lib.mmw_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr + byte_offset, out.nbytes))
        return out

    def download_widened(self, shape, dtype, byte_offset: int = 0, staging=None) -> np.ndarray:
        """``download(shape, dtype).astype(wider)`` with the widening done on the device (float32 -> float64, complex64 ->
        complex128) into ``staging`` (a DeviceBuffer of twice the bytes): for the multi-megabyte arrays the reference's
        dtypes oblige, one host core converting costs more than twice the bytes over PCIe."""
        narrow = np.dtype(dtype)
        wide = np.dtype(np.complex128 if narrow == np.complex64 else np.float64)
        out = self.ctx.result_array(shape, wide)
        n_f32 = out.size * (2 if narrow == np.complex64 else 1)
        if byte_offset + n_f32 * 4 > self.nbytes or staging is None or staging.nbytes < out.nbytes:
            raise ValueError("download_widened: source or staging buffer too small")
        check(self.ctx.lib.mmw_widen_f32_f64(self.ctx.handle, self.ptr + byte_offset, staging.ptr, n_f32))
        check(self.ctx.lib.mmw_memcpy_d2h(self.ctx.handle, out.ctypes.data, staging.ptr, out.nbytes))
        return out

    def zero(self):
        check(self.ctx.lib.mmw_memset(self.ctx.handle, self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr:
            check(self.ctx.lib.mmw_free(self.ctx.handle, self.ptr))
            self.ptr = 0


class Context:
    """One HIP device + stream.  Not thread-safe; one per host thread (include/mmwgpu.h)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        if device_count() < 1:
            raise MmwGpuError("no HIP device visible: the MI355X HIP path is the only backend")
        h = C.c_void_p()
        check(self.lib.mmw_ctx_create(C.byref(h), int(device)))
        self.handle = h
        self.device = int(device)
        self._cache = {}
        self._result_pool = None        # pinned result blocks by size class (the process-wide default context only)
        self._result_bytes = 0

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def cached(self, key, nbytes: int) -> DeviceBuffer:
        """Grow-only named buffer, reused across process() calls of one processor."""
        buf = self._cache.get(key)
        if buf is None or buf.nbytes < nbytes:
            if buf is not None:
                buf.free()
            buf = self.alloc(nbytes)
            self._cache[key] = buf
        return buf

    def sync(self):
        check(self.lib.mmw_sync(self.handle))

    # host streaming ------------------------------------------------------
    def set_option(self, name: str, value=None) -> None:
        """A tuning / test switch of this context (``mmw_diag_set_option``; INTEGRATION.md lists the names).  ``None`` removes
        the context's value again (the environment / the default applies)."""
        check(self.lib.mmw_diag_set_option(self.handle, name.encode(), -2**31 if value is None else int(value)))

    def result_array(self, shape, dtype) -> np.ndarray:
        """A fresh, caller-owned ndarray for a download.  On the process-wide default context, results of 1 MiB and more come
        out of a pool of PINNED blocks: a device-to-host copy into pageable memory the caller has never touched pays page
        faults and a bounce buffer (the 33 MB complex128 cube of ``compute_3d_windowed_fft``: 3.4 ms; pinned: the PCIe time).
        A block goes back to the pool when the last view of the array is gone (the array's base object carries a
        finalizer), so a frame loop that drops or overwrites its results allocates nothing after the first frames.  Contexts
        that can be closed while their results are alive, small results and a pool grown past RESULT_POOL_CAP bytes use
        ``np.empty``."""
        dt = np.dtype(dtype)
        count = int(np.prod(shape))
        nbytes = count * dt.itemsize
        if self._result_pool is None or nbytes < (1 << 20):
            return np.empty(shape, dtype=dt)
        size = 1 << (nbytes - 1).bit_length()
        free = self._result_pool.setdefault(size, [])
        if free:
            ptr = free.pop()
        else:
            if self._result_bytes + size > RESULT_POOL_CAP:
                # make room out of idle blocks of other sizes (a one-off large download must not end the pooling of the rest)
                for other in sorted(self._result_pool, reverse=True):
                    idle = self._result_pool[other]
                    while idle and self._result_bytes + size > RESULT_POOL_CAP:
                        check(self.lib.mmw_host_free(self.handle, C.c_void_p(idle.pop())))
                        self._result_bytes -= other
                if self._result_bytes + size > RESULT_POOL_CAP:
                    return np.empty(shape, dtype=dt)
            p = C.c_void_p()
            check(self.lib.mmw_host_alloc(self.handle, C.byref(p), size))
            ptr = p.value
            self._result_bytes += size
        buf = (C.c_char * size).from_address(ptr)
        weakref.finalize(buf, free.append, ptr).atexit = False
        return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)

    def host_array(self, shape, dtype) -> np.ndarray:
        """Pinned host memory as an ndarray (freed with the context)."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        check(self.lib.mmw_host_alloc(self.handle, C.byref(p), nbytes))
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.__array_interface__["data"][0]] = p.value
        return arr

    def host_free(self, arr: np.ndarray) -> None:
        """Release a ``host_array`` block now (the caller drops every view of it); blocks never freed go with the context."""
        p = getattr(self, "_pinned", {}).pop(arr.__array_interface__["data"][0], None)
        if p is None:
            raise ValueError("not a block of this context's host_array()")
        check(self.lib.mmw_host_free(self.handle, C.c_void_p(p)))

    def event(self):
        e = C.c_void_p()
        check(self.lib.mmw_event_create(self.handle, C.byref(e)))
        return e

    def copy_async(self, dst, src, nbytes, to_host=False, queue=QUEUE_COPY):
        check(self.lib.mmw_memcpy_async(self.handle, dst, src, int(nbytes), int(bool(to_host)), int(queue)))

    def record(self, event, queue):
        check(self.lib.mmw_event_record(self.handle, event, int(queue)))

    def wait(self, queue, event):
        check(self.lib.mmw_queue_wait_event(self.handle, int(queue), event))

    def event_sync(self, event):
        check(self.lib.mmw_event_sync(self.handle, event))

    def timer_start(self):
        check(self.lib.mmw_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        check(self.lib.mmw_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        """on: False/0 off, True/1 every launch group, n > 1 every n-th launch group of each family."""
        check(self.lib.mmw_profile_enable(self.handle, int(on)))

    def profile_reset(self):
        check(self.lib.mmw_profile_reset(self.handle))

    def profile_get(self, family: str):
        ms, n = C.c_float(0), C.c_int(0)
        check(self.lib.mmw_profile_get(self.handle, family.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if self.handle:
            self.lib.mmw_ctx_destroy(self.handle)
            self.handle = None
            self._cache = {}

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass


class BufferSet:
    """Grow-only named device buffers owned by one processor / detector object."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._bufs = {}

    def get(self, name: str, nbytes: int) -> DeviceBuffer:
        buf = self._bufs.get(name)
        if buf is None or buf.nbytes < nbytes:
            if buf is not None:
                buf.free()
            buf = self.ctx.alloc(max(int(nbytes), 16))
            self._bufs[name] = buf
        return buf

    def free(self):
        for buf in self._bufs.values():
            try:
                buf.free()
            except Exception:
                pass
        self._bufs = {}

    def __del__(self):
        if getattr(self.ctx, "handle", None):
            self.free()


_default_ctx = None


def default_context() -> Context:
    """Process-wide context on ``$MMW_DEVICE`` (else ``$LOCAL_RANK``, else 0)."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("MMW_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = device_count()
        if n < 1:
            raise MmwGpuError("no HIP device visible: the MI355X HIP path is the only backend")
        _default_ctx = Context(dev % n)
        if os.environ.get("MMW_PINNED_RESULTS", "1") != "0":
            _default_ctx._result_pool = {}
    return _default_ctx


def int_array(values):
    vals = [int(v) for v in values]
    return (C.c_int * max(1, len(vals)))(*vals), len(vals)
