"""The handful of HIP runtime calls the multi-GPU host layer needs (device / pinned buffers, 2-D async copies,
stream synchronisation), bound with ctypes on the SAME runtime copy libbskgpu.so uses (``_lib.load()`` pins it).
No torch on this path."""
import ctypes as C

from . import _lib

hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice = 1, 2, 3
_RT = None


class HipError(RuntimeError):
    pass


def runtime():
    global _RT
    if _RT is not None:
        return _RT
    _lib.load()                                # makes sure the process-wide runtime choice has been made
    cand = _lib._share_hip_runtime_with_torch() or "libamdhip64.so"
    try:
        rt = C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        rt = C.CDLL("/opt/rocm/lib/libamdhip64.so", mode=C.RTLD_GLOBAL)
    vp, sz = C.c_void_p, C.c_size_t
    rt.hipGetErrorString.restype = C.c_char_p
    rt.hipGetErrorString.argtypes = [C.c_int]
    rt.hipSetDevice.argtypes = [C.c_int]
    rt.hipGetDevice.argtypes = [C.POINTER(C.c_int)]
    rt.hipGetDeviceCount.argtypes = [C.POINTER(C.c_int)]
    rt.hipMalloc.argtypes = [C.POINTER(vp), sz]
    rt.hipFree.argtypes = [vp]
    rt.hipHostMalloc.argtypes = [C.POINTER(vp), sz, C.c_uint]
    rt.hipHostFree.argtypes = [vp]
    rt.hipMemcpyAsync.argtypes = [vp, vp, sz, C.c_int, vp]
    rt.hipMemcpy2DAsync.argtypes = [vp, sz, vp, sz, sz, sz, C.c_int, vp]
    rt.hipStreamSynchronize.argtypes = [vp]
    rt.hipMemsetAsync.argtypes = [vp, C.c_int, sz, vp]
    _RT = rt
    return rt


def check(rc, what="hip"):
    if rc != 0:
        msg = runtime().hipGetErrorString(rc)
        raise HipError("%s: %s (%d)" % (what, msg.decode() if msg else "?", rc))


def device_count():
    n = C.c_int(0)
    rc = runtime().hipGetDeviceCount(C.byref(n))
    return n.value if rc == 0 else 0


def current_device():
    d = C.c_int(-1)
    check(runtime().hipGetDevice(C.byref(d)), "hipGetDevice")
    return d.value


class device_guard(object):
    """``with device_guard(d): ...`` - the calling thread's current HIP device is ``d`` inside and whatever it was before
    afterwards (the C library's DeviceGuard, for the handful of runtime calls this package makes from Python).  A process
    that shards a batch over several GPUs usually also runs a torch policy: that one must keep finding ITS device current."""

    def __init__(self, device):
        self.device, self.prev = int(device), None

    def __enter__(self):
        self.prev = current_device()
        if self.prev != self.device:
            set_device(self.device)
        return self

    def __exit__(self, *exc):
        if self.prev is not None and self.prev != self.device:
            set_device(self.prev)
        return False


class DeviceBuffer(object):
    """hipMalloc'ed bytes on ``device``; exposes ``__cuda_array_interface__`` through ``view``."""

    def __init__(self, nbytes, device):
        rt = runtime()
        p = C.c_void_p()
        with device_guard(device):
            check(rt.hipMalloc(C.byref(p), int(nbytes)), "hipMalloc")
        self.ptr, self.nbytes, self.device = p.value, int(nbytes), int(device)

    def free(self):
        if self.ptr:
            runtime().hipFree(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedBuffer(object):
    """Page-locked host bytes with a numpy view (D2H targets of the per-device copies)."""

    def __init__(self, nbytes):
        import numpy as np
        p = C.c_void_p()
        check(runtime().hipHostMalloc(C.byref(p), int(nbytes), 0), "hipHostMalloc")
        self.ptr, self.nbytes = p.value, int(nbytes)
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self.ptr), dtype=np.uint8)

    def free(self):
        if self.ptr:
            self.array = None
            runtime().hipHostFree(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def memcpy2d_async(dst, dpitch, src, spitch, width, height, kind, stream):
    check(runtime().hipMemcpy2DAsync(C.c_void_p(dst), dpitch, C.c_void_p(src), spitch, width, height, kind,
                                     C.c_void_p(stream)), "hipMemcpy2DAsync")


def stream_sync(stream):
    check(runtime().hipStreamSynchronize(C.c_void_p(stream)), "hipStreamSynchronize")


def set_device(device):
    check(runtime().hipSetDevice(int(device)), "hipSetDevice")
