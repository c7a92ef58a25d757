"""Direct librccl.so binding for the ONE exchange step of the path: delivering the observation batch.

Spacecraft never interact (one ``scObject`` per simulator, reference simulators/leoPowerAttitudeSimulator.py:213),
so stepping has no collective.  What moves between GPUs is only the per-GPU ``f64[5][n_local]`` observation shard
(+ reward, reason) when ONE consumer wants the whole batch on one device (BASELINE configs[3]).
xGMI is point-to-point — every GPU has its own link to the root — so the gather is written as what the fabric is:
a grouped set of ``ncclSend`` / ``ncclRecv`` pairs straight from the library's SoA rows into the root's
``[5][n_total]`` buffer at the shard's column offset (7 inbound links busy at once, no ring, no staging copy, no
concatenation afterwards), enqueued on the propagator handles' OWN streams, so the exchange is ordered after the
step kernel without any host synchronisation.

Two ways to get communicators: ``init_all(devices)`` — one process driving several GPUs (ShardedPropagator) — and
``init_rank(world, rank, uid)`` — one process per GPU, the unique id distributed by whatever the caller has
(bench.py: one ``torch.distributed`` object broadcast).  No torch in here.
"""
import ctypes as C
import os

from . import _hip, _lib

ncclUint8, ncclInt32, ncclFloat64 = 1, 2, 8
ncclSum = 0
_RCCL = None


class RcclError(RuntimeError):
    pass


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _candidates():
    out = []
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            # the copy torch.distributed's "nccl" backend uses: one RCCL per process, like the HIP runtime
            out.append(os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so"))
    except Exception:
        pass
    out += ["librccl.so", "/opt/rocm/lib/librccl.so", "librccl.so.1"]
    return out


def load():
    global _RCCL
    if _RCCL is not None:
        return _RCCL
    _lib.load()          # pins the process's HIP runtime first
    err = None
    for cand in _candidates():
        if os.path.isabs(cand) and not os.path.exists(cand):
            continue
        try:
            lib = C.CDLL(cand, mode=C.RTLD_GLOBAL)
            break
        except OSError as e:
            err = e
    else:
        raise RcclError("librccl.so not loadable: %s" % err)
    vp, sz = C.c_void_p, C.c_size_t
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, _UniqueId, C.c_int]
    lib.ncclCommInitAll.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
    lib.ncclCommDestroy.argtypes = [vp]
    lib.ncclGroupStart.argtypes = []
    lib.ncclGroupEnd.argtypes = []
    lib.ncclSend.argtypes = [vp, sz, C.c_int, C.c_int, vp, vp]
    lib.ncclRecv.argtypes = [vp, sz, C.c_int, C.c_int, vp, vp]
    lib.ncclAllGather.argtypes = [vp, vp, sz, C.c_int, vp, vp]
    lib.ncclAllReduce.argtypes = [vp, vp, sz, C.c_int, C.c_int, vp, vp]
    _RCCL = lib
    return lib


def _ck(rc, what):
    if rc != 0:
        msg = load().ncclGetErrorString(rc)
        raise RcclError("%s: %s (%d)" % (what, msg.decode() if msg else "?", rc))


def unique_id():
    """128 opaque bytes; rank 0 makes them, every rank of the communicator needs the same ones."""
    uid = _UniqueId()
    _ck(load().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return C.string_at(C.addressof(uid), 128)


class Comm(object):
    """One rank of a communicator (ncclComm_t) bound to ``device``."""

    def __init__(self, handle, rank, world, device):
        self.handle, self.rank, self.world, self.device = handle, int(rank), int(world), int(device)

    @classmethod
    def init_rank(cls, world, rank, uid_bytes, device):
        lib = load()
        if len(uid_bytes) != 128:
            raise ValueError("unique id must be 128 bytes")
        uid = _UniqueId()
        C.memmove(C.addressof(uid), uid_bytes, 128)
        h = C.c_void_p()
        with _hip.device_guard(device):       # the communicator binds the device current at this call; the caller's stays
            _ck(lib.ncclCommInitRank(C.byref(h), int(world), uid, int(rank)), "ncclCommInitRank")
        return cls(h.value, rank, world, device)

    @classmethod
    def init_all(cls, devices):
        """Communicator over ``devices`` (distinct GPUs) driven by THIS process: one Comm per device, rank = position."""
        lib = load()
        devices = [int(d) for d in devices]
        if len(set(devices)) != len(devices):
            raise ValueError("RCCL needs distinct devices in one communicator, got %r" % (devices,))
        n = len(devices)
        hs = (C.c_void_p * n)()
        _ck(lib.ncclCommInitAll(hs, n, (C.c_int * n)(*devices)), "ncclCommInitAll")
        return [cls(hs[r], r, n, devices[r]) for r in range(n)]

    def destroy(self):
        if self.handle:
            load().ncclCommDestroy(C.c_void_p(self.handle))
            self.handle = None


def group_start():
    _ck(load().ncclGroupStart(), "ncclGroupStart")


def group_end():
    _ck(load().ncclGroupEnd(), "ncclGroupEnd")


def column_offsets(sizes):
    offs, acc = [], 0
    for s in sizes:
        offs.append(acc)
        acc += int(s)
    return offs, acc


class GatherBuf(object):
    """One buffer of the gather: this rank's ``[rows][n_r]`` shard at ``src_ptr`` (row pitch ``src_pitch_bytes``) lands in
    the root's ``[rows][n_total]`` buffer at ``out_ptr`` (only read on the root) at the rank's column offset."""

    def __init__(self, src_ptr, src_pitch_bytes, rows, out_ptr, itemsize=8, dtype=ncclFloat64):
        self.src_ptr, self.src_pitch_bytes, self.rows, self.out_ptr = int(src_ptr), int(src_pitch_bytes), int(rows), int(out_ptr or 0)
        self.itemsize, self.dtype = int(itemsize), int(dtype)


def step_output_bufs(obs_ptr, obs_pitch_bytes, reward_ptr, reason_ptr, out_obs, out_reward, out_reason):
    """What a consumer on the root GPU needs from every shard to train on ("obs f64[5][N/8]
    (+ reward, done mask)"): five observation rows, one reward row, one reason row (uint8; done = reason != 0) - seven
    rows per rank, moved in ONE group."""
    return [GatherBuf(obs_ptr, obs_pitch_bytes, 5, out_obs), GatherBuf(reward_ptr, 0, 1, out_reward),
            GatherBuf(reason_ptr, 0, 1, out_reason, itemsize=1, dtype=ncclUint8)]


def gather_bytes(sizes, bufs, root):
    """Bytes that cross the fabric per gather (everything but the root's own shard)."""
    return sum(int(n) for r, n in enumerate(sizes) if r != root) * sum(b.rows * b.itemsize for b in bufs)


def enqueue_gather(comm, stream, root, sizes, bufs):
    """This rank's part of the direct gather of several buffers (``GatherBuf``) in one group: column offset of rank r = sum
    of the sizes before it.  MUST be called between group_start() and group_end(), on every rank of the communicator (one
    process per rank: once; one process for all: once per Comm).  Non-root ranks send ``rows`` messages of ``sizes[rank]``
    items per buffer, the root posts the matching receives (same order: messages between two ranks pair up in posting order)
    and copies its own shard device-to-device on the same stream after the group (copy_own)."""
    lib = load()
    offs, n_total = column_offsets(sizes)
    vp = C.c_void_p
    if comm.rank == root:
        for r in range(comm.world):
            if r == root or sizes[r] == 0:
                continue
            for b in bufs:
                for f in range(b.rows):
                    _ck(lib.ncclRecv(vp(b.out_ptr + (f * n_total + offs[r]) * b.itemsize), int(sizes[r]), b.dtype, r, vp(comm.handle), vp(stream)),
                        "ncclRecv")
    elif sizes[comm.rank]:
        for b in bufs:
            for f in range(b.rows):
                _ck(lib.ncclSend(vp(b.src_ptr + f * b.src_pitch_bytes), int(sizes[comm.rank]), b.dtype, root, vp(comm.handle), vp(stream)), "ncclSend")


def copy_own(comm, stream, root, sizes, bufs):
    """The root's own shard: one strided device-to-device copy per buffer on its stream (call after group_end())."""
    if comm.rank != root or not sizes[root]:
        return
    offs, n_total = column_offsets(sizes)
    with _hip.device_guard(comm.device):
        for b in bufs:
            _hip.memcpy2d_async(b.out_ptr + offs[root] * b.itemsize, n_total * b.itemsize, b.src_ptr, b.src_pitch_bytes or int(sizes[root]) * b.itemsize,
                                int(sizes[root]) * b.itemsize, b.rows, _hip.hipMemcpyDeviceToDevice, stream)


# ---------------------------------------------------------------------------------------------------------------------------
# Rank-major root layout: ONE message per rank and data type.  The column-offset form above lands every row of every shard at
# its place in a [rows][n_total] buffer, which takes `rows` messages per rank and buffer (7 x 7 = 49 receives on the root of an
# 8-GPU node).  Here the root's buffer is laid out BY RANK - rank r's block is f64[6][n_r] (five observation rows, then the
# reward row) at byte offset 48 * offs[r], and the done reasons u8[n_total] in rank (= env-index) order behind the f64 part -
# and the library keeps observation rows and reward row in ONE allocation f64[6][stride] (bsk_create), so a shard whose size
# equals its stride (n_r a multiple of 256: BASELINE configs[3]'s 131 072 and configs[2]'s 65 536 are) is one contiguous
# 6 n_r doubles: TWO messages per rank (f64 block, u8 reasons) instead of seven, 14 receives on the root instead of 49.
# A shard with padding (n_r not a multiple of 256, or a library built with another row pitch) sends its six rows one by one into
# the same block (seven messages, as before).  Which of the two a rank does is decided ONCE, by the sender from its real buffers
# (RankMajorBufs.contiguous: pitch and reward address), and exchanged when the gather object is built (`contig`, one flag per rank:
# DirectRcclGather.__init__ all-gathers them), so that sender and receiver post matching messages whatever the pitch is and nothing
# has to be refused between group_start() and group_end().
STRIDE_QUANTUM = 256          # bsk_create pads the env stride to a multiple of this


def rank_major_contiguous(n_r):
    """What a shard of n_r envs is with the library's default row pitch (the exchanged flags are authoritative)."""
    return int(n_r) > 0 and int(n_r) % STRIDE_QUANTUM == 0


def rank_major_flags(sizes, contig=None):
    return [bool(c) and int(n) > 0 for n, c in zip(sizes, contig)] if contig is not None else [rank_major_contiguous(n) for n in sizes]


class RankMajorBufs(object):
    """This rank's sources - obs ``f64[5][pitch]``, reward ``f64[n_r]`` (directly behind the observation rows when the shard is
    contiguous), reason ``u8[n_r]`` - and the root's rank-major destination: ``out_f64`` (6 * n_total doubles), ``out_u8``
    (n_total bytes); the ``out_*`` pointers are only read on the root."""

    def __init__(self, obs_ptr, obs_pitch_bytes, reward_ptr, reason_ptr, out_f64, out_u8):
        self.obs_ptr, self.obs_pitch_bytes, self.reward_ptr, self.reason_ptr = int(obs_ptr), int(obs_pitch_bytes), int(reward_ptr), int(reason_ptr)
        self.out_f64, self.out_u8 = int(out_f64 or 0), int(out_u8 or 0)

    def contiguous(self, n_r):
        """The six f64 rows of this shard are one block (the precondition of the single-message form)."""
        return self.obs_pitch_bytes == 8 * int(n_r) and self.reward_ptr == self.obs_ptr + 5 * self.obs_pitch_bytes


def rank_major_messages(sizes, root, contig=None):
    """Receives the root posts per gather (= sends of all other ranks together)."""
    flags = rank_major_flags(sizes, contig)
    return sum((2 if flags[r] else 7) for r, n in enumerate(sizes) if r != root and n)


def enqueue_gather_rank_major(comm, stream, root, sizes, b, contig=None):
    """This rank's part of the rank-major gather (between group_start() and group_end(), every rank of the communicator).
    ``contig``: one flag per rank, exchanged beforehand - rank r sends its f64 part as one block; None: the default pitch's rule.
    A sender whose buffers do not bear out its flag is refused BEFORE any message is posted (nothing raises inside an open group
    once the flags come from RankMajorBufs.contiguous itself)."""
    lib = load()
    offs, n_total = column_offsets(sizes)
    vp = C.c_void_p
    flags = rank_major_flags(sizes, contig)
    if comm.rank == root:
        for r in range(comm.world):
            n_r = int(sizes[r])
            if r == root or n_r == 0:
                continue
            blk = b.out_f64 + 48 * offs[r]
            if flags[r]:
                _ck(lib.ncclRecv(vp(blk), 6 * n_r, ncclFloat64, r, vp(comm.handle), vp(stream)), "ncclRecv")
            else:
                for f in range(6):
                    _ck(lib.ncclRecv(vp(blk + 8 * f * n_r), n_r, ncclFloat64, r, vp(comm.handle), vp(stream)), "ncclRecv")
            _ck(lib.ncclRecv(vp(b.out_u8 + offs[r]), n_r, ncclUint8, r, vp(comm.handle), vp(stream)), "ncclRecv")
        return
    n_r = int(sizes[comm.rank])
    if n_r == 0:
        return
    if flags[comm.rank]:
        if not b.contiguous(n_r):
            raise RcclError("rank-major gather: a shard of %d envs must be one f64[6][%d] block (observation pitch %d bytes, reward at +%d)"
                            % (n_r, n_r, b.obs_pitch_bytes, b.reward_ptr - b.obs_ptr))
        _ck(lib.ncclSend(vp(b.obs_ptr), 6 * n_r, ncclFloat64, root, vp(comm.handle), vp(stream)), "ncclSend")
    else:
        for f in range(5):
            _ck(lib.ncclSend(vp(b.obs_ptr + f * b.obs_pitch_bytes), n_r, ncclFloat64, root, vp(comm.handle), vp(stream)), "ncclSend")
        _ck(lib.ncclSend(vp(b.reward_ptr), n_r, ncclFloat64, root, vp(comm.handle), vp(stream)), "ncclSend")
    _ck(lib.ncclSend(vp(b.reason_ptr), n_r, ncclUint8, root, vp(comm.handle), vp(stream)), "ncclSend")


def copy_own_rank_major(comm, stream, root, sizes, b):
    """The root's own shard into its block (after group_end()): observation rows (strided), reward row, reasons."""
    n_r = int(sizes[root])
    if comm.rank != root or not n_r:
        return
    offs, _ = column_offsets(sizes)
    blk = b.out_f64 + 48 * offs[root]
    with _hip.device_guard(comm.device):
        _hip.memcpy2d_async(blk, 8 * n_r, b.obs_ptr, b.obs_pitch_bytes, 8 * n_r, 5, _hip.hipMemcpyDeviceToDevice, stream)
        _hip.memcpy2d_async(blk + 40 * n_r, 8 * n_r, b.reward_ptr, 8 * n_r, 8 * n_r, 1, _hip.hipMemcpyDeviceToDevice, stream)
        _hip.memcpy2d_async(b.out_u8 + offs[root], n_r, b.reason_ptr, n_r, n_r, 1, _hip.hipMemcpyDeviceToDevice, stream)


def comm_count(comm):
    """ncclCommCount: how many ranks this communicator really spans (what a first multi-GPU run reads back as proof)."""
    lib = load()
    n = C.c_int(-1)
    lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    _ck(lib.ncclCommCount(C.c_void_p(comm.handle), C.byref(n)), "ncclCommCount")
    return n.value


def enqueue_gather_rows(comm, stream, root, sizes, src_ptr, src_pitch_bytes, rows, out_ptr, itemsize=8, dtype=ncclFloat64):
    """One buffer (the observation rows alone): see enqueue_gather."""
    enqueue_gather(comm, stream, root, sizes, [GatherBuf(src_ptr, src_pitch_bytes, rows, out_ptr, itemsize, dtype)])


def copy_own_rows(comm, stream, root, sizes, src_ptr, src_pitch_bytes, rows, out_ptr, itemsize=8):
    copy_own(comm, stream, root, sizes, [GatherBuf(src_ptr, src_pitch_bytes, rows, out_ptr, itemsize)])


def all_reduce_sum_f64(comm, stream, send_ptr, recv_ptr, count):
    """ncclAllReduce(sum) of ``count`` doubles on ``stream``: the batch scalars {sum of rewards, number of done envs} of a
    sharded step - two doubles per rank.  Callers reduce OUT OF PLACE: the send buffer is the handle's own
    bsk_get_batch_stats_device() block, which the library only refreshes after the next step - reduced in place, a second
    call between two steps would sum the already-summed values (world x the batch sum)."""
    _ck(load().ncclAllReduce(C.c_void_p(send_ptr), C.c_void_p(recv_ptr), int(count), ncclFloat64, ncclSum, C.c_void_p(comm.handle), C.c_void_p(stream)),
        "ncclAllReduce")
