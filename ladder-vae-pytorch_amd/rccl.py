"""The gradient exchange's communicator: a thin binding of liblvae_hip.so's lvae_allreduce_* entry points (csrc/allreduce.hip), which own
a private RCCL communicator and the fork / join of the side stream around every bucket.

Why a communicator of our own instead of torch.distributed's collectives: ProcessGroupNCCL keeps a Work object, events and a polling
watchdog per collective, and events recorded while a stream is capturing must not be queried — round 3 saw the watchdog abort the process
with "operation not permitted on an event last recorded in a capturing stream" on exactly this path (a race: most runs pass). An
ncclAllReduce on our own communicator leaves nothing for a watchdog to look at: it is a plain launch on the stream we pass, captured like
any other kernel. The process group is still what hands out the communicator id, broadcasts the initial parameters and runs barriers /
timing reductions, all eagerly and outside the step.

librccl.so is the one torch itself loaded (torch/lib/librccl.so): backend "nccl" IS RCCL on ROCm, over xGMI between the GPUs of a node.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _C


class RcclError(RuntimeError):
    pass


def librccl_path():
    return os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')


def _call(name, *args):
    try:
        _C.call(name, *args)
    except _C.LvaeHipError as e:
        raise RcclError(str(e)) from None


class Comm:
    """One communicator over the ranks of `group` (default: the world). The id is created by rank 0 and handed out through the process
    group (an eager broadcast of 128 bytes). The current device must be this rank's GPU. ncclCommInitRank is a collective: every rank of
    the group must construct its Comm."""

    def __init__(self, group=None):
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        path = librccl_path().encode()
        uid = (C.c_ubyte * 128)()
        if self.rank == 0:
            _call('lvae_allreduce_unique_id', path, C.cast(uid, C.c_void_p))
        dev = torch.device('cuda', torch.cuda.current_device())
        t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(dev)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(t, src=src, group=group)
        raw = (C.c_ubyte * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
        self.handle = C.c_void_p()
        _call('lvae_allreduce_init', path, C.cast(raw, C.c_void_p), self.world, self.rank, C.byref(self.handle))

    @staticmethod
    def _check(tensor):
        if tensor.dtype != torch.float32 or not tensor.is_contiguous() or not tensor.is_cuda:
            raise RcclError('the exchange takes a contiguous float32 device tensor')

    def enqueue(self, tensor, launch_stream, side_stream, scratch=None):
        """In-place float32 SUM over the ranks on `side_stream`, behind everything issued on `launch_stream` so far (the library forks).
        scratch (same size): the out-of-place form for one-rank rehearsals (an in-place all-reduce of one rank enqueues nothing)."""
        self._check(tensor)
        if scratch is not None and (scratch.numel() < tensor.numel() or scratch.dtype != torch.float32 or not scratch.is_cuda):
            raise RcclError('scratch must be a float32 device tensor at least as large as the bucket')
        _call('lvae_allreduce_enqueue', self.handle, tensor.data_ptr(), tensor.numel(), scratch.data_ptr() if scratch is not None else None,
              launch_stream.cuda_stream, side_stream.cuda_stream)

    def wait(self, launch_stream, side_stream):
        """`launch_stream` waits for everything enqueued on `side_stream` so far."""
        _call('lvae_allreduce_wait', self.handle, launch_stream.cuda_stream, side_stream.cuda_stream)

    def all_reduce_(self, tensor, stream=None, scratch=None):
        """The same exchange on ONE stream (default: torch's current stream), no fork: a plain launch, capturable."""
        st = stream or torch.cuda.current_stream(tensor.device)
        self.enqueue(tensor, st, st, scratch)

    def destroy(self):
        if self.handle:
            _C.load().lvae_allreduce_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()

    def __del__(self):   # a communicator nobody closed must not leak its RCCL comm and two events
        try:
            self.destroy()
        except Exception:  # noqa: BLE001  (interpreter shutdown: the library may be gone already)
            pass
