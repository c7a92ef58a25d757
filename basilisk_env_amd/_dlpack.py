"""DLPack export of library-owned device buffers, written against the DLPack C structs with ctypes only
(no torch / numpy on this path): ``from_dlpack(view)`` in any consumer gives a zero-copy tensor over the
HBM buffer libbskgpu.so owns ("torch/DLPack zero-copy observation hand-off").

Device type is kDLROCM (10); a ROCm build of PyTorch maps it to its ``cuda`` device.
"""
import ctypes as C

kDLROCM = 10
_CODES = {"f": 2, "i": 0, "u": 1, "b": 6}   # kDLFloat, kDLInt, kDLUInt, kDLBool


class DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int), ("device_id", C.c_int)]


class DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", DLDevice), ("ndim", C.c_int), ("dtype", DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class DLManagedTensor(C.Structure):
    pass


_DELETER = C.CFUNCTYPE(None, C.POINTER(DLManagedTensor))
DLManagedTensor._fields_ = [("dl_tensor", DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DELETER)]

# everything a live export needs (struct, shape / stride arrays, the Python owner of the buffer), keyed by the
# struct's address; released by the consumer through `deleter`, or by the capsule if nobody consumed it
_LIVE = {}


@_DELETER
def _deleter(mt):
    _LIVE.pop(C.addressof(mt.contents), None)


_CAPSULE_DTOR = C.CFUNCTYPE(None, C.c_void_p)
_api = C.pythonapi
_api.PyCapsule_New.restype = C.py_object
_api.PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, _CAPSULE_DTOR]
_api.PyCapsule_IsValid.restype = C.c_int
_api.PyCapsule_IsValid.argtypes = [C.c_void_p, C.c_char_p]
_api.PyCapsule_GetPointer.restype = C.c_void_p
_api.PyCapsule_GetPointer.argtypes = [C.c_void_p, C.c_char_p]
_NAME = b"dltensor"


@_CAPSULE_DTOR
def _capsule_dtor(cap):
    # a consumer renames the capsule to "used_dltensor" and owns the deleter call; an unconsumed one is ours
    if _api.PyCapsule_IsValid(cap, _NAME):
        _LIVE.pop(_api.PyCapsule_GetPointer(cap, _NAME), None)


def typestr_to_dl(typestr):
    """numpy-style typestr ('<f8', '<u8', '|u1', '<i4', '|b1') -> DLDataType."""
    kind, size = typestr[1], int(typestr[2:])
    return DLDataType(_CODES[kind], 8 * size, 1)


def make_capsule(ptr, shape, typestr, strides_bytes=None, device_id=0, owner=None, device_type=kDLROCM):
    """PyCapsule "dltensor" over device memory at ``ptr``.  ``strides_bytes`` as in the array interfaces
    (None = C-contiguous); DLPack counts strides in elements."""
    itemsize = int(typestr[2:])
    nd = len(shape)
    shp = (C.c_int64 * nd)(*[int(s) for s in shape])
    if strides_bytes is None:
        st, acc = [0] * nd, 1
        for k in range(nd - 1, -1, -1):
            st[k] = acc
            acc *= int(shape[k])
    else:
        if any(int(b) % itemsize for b in strides_bytes):
            raise ValueError("strides must be multiples of the item size")
        st = [int(b) // itemsize for b in strides_bytes]
    strd = (C.c_int64 * nd)(*st)
    mt = DLManagedTensor()
    mt.dl_tensor.data = int(ptr)
    mt.dl_tensor.device = DLDevice(int(device_type), int(device_id))
    mt.dl_tensor.ndim = nd
    mt.dl_tensor.dtype = typestr_to_dl(typestr)
    mt.dl_tensor.shape = shp
    mt.dl_tensor.strides = strd
    mt.dl_tensor.byte_offset = 0
    mt.manager_ctx = None
    mt.deleter = _deleter
    addr = C.addressof(mt)
    _LIVE[addr] = (mt, shp, strd, owner)
    return _api.PyCapsule_New(addr, _NAME, _capsule_dtor)


def live_exports():
    """Number of exported tensors a consumer still holds (tests)."""
    return len(_LIVE)
