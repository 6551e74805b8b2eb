"""A private RCCL communicator for the gradient exchange INSIDE the captured step graph.

torch.distributed's ProcessGroupNCCL works for eager collectives, but its bookkeeping is not made for collectives that are captured
into a hipGraph on a side stream: every collective creates a Work object with events, a watchdog thread polls them, and events recorded
while a stream is capturing must not be queried — round 3 saw the watchdog abort the process with "operation not permitted on an event
last recorded in a capturing stream" on exactly this path (a race: most runs pass). Calling ncclAllReduce ourselves on our own
communicator leaves nothing for a watchdog to look at: the call is a plain kernel launch on the stream we pass, captured like any
other kernel. (The same reasoning as vLLM's pynccl wrapper.) The process group is still what exchanges the communicator id, broadcasts
the initial parameters and runs barriers / timing reductions, all eagerly and outside the step.

librccl.so is the one torch itself loaded (torch/lib/librccl.so): `backend "nccl" IS RCCL on ROCm`, over xGMI between the GPUs of a node.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_FLOAT32, NCCL_SUM = 7, 0   # rccl.h: ncclDataType_t / ncclRedOp_t


class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_byte * 128)]   # NCCL_UNIQUE_ID_BYTES


_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
        lib = C.CDLL(path)
        lib.ncclGetUniqueId.restype = C.c_int
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.restype = C.c_int
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclAllReduce.restype = C.c_int
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.restype = C.c_int
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        _lib = lib
    return _lib


class RcclError(RuntimeError):
    pass


def _chk(rc, what):
    if rc != 0:
        raise RcclError('%s failed: %s' % (what, _load().ncclGetErrorString(rc).decode()))


class Comm:
    """One communicator over the ranks of `group` (default: the world). The id is created by rank 0 and handed out through the
    process group (an eager broadcast of 128 bytes). The current device must be this rank's GPU."""

    def __init__(self, group=None):
        lib = _load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        uid = _UniqueId()
        if self.rank == 0:
            _chk(lib.ncclGetUniqueId(C.byref(uid)), 'ncclGetUniqueId')
        dev = torch.device('cuda', torch.cuda.current_device())
        t = torch.frombuffer(bytearray(bytes(uid.internal)), dtype=torch.uint8).to(dev)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(t, src=src, group=group)
        raw = bytes(t.cpu().numpy().tobytes())
        C.memmove(C.byref(uid), raw, 128)
        self.comm = C.c_void_p()
        _chk(lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), 'ncclCommInitRank')

    def all_reduce_(self, tensor, stream=None):
        """In-place float32 SUM over the ranks, enqueued on `stream` (default: torch's current stream). A plain launch: capturable."""
        if tensor.dtype != torch.float32 or not tensor.is_contiguous() or not tensor.is_cuda:
            raise RcclError('all_reduce_ takes a contiguous float32 device tensor')
        st = (stream or torch.cuda.current_stream(tensor.device)).cuda_stream
        _chk(_load().ncclAllReduce(tensor.data_ptr(), tensor.data_ptr(), tensor.numel(), NCCL_FLOAT32, NCCL_SUM, self.comm, st), 'ncclAllReduce')

    def destroy(self):
        if self.comm:
            _load().ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
