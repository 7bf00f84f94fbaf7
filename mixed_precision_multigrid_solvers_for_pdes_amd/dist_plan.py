"""Cycle plans of the decomposed solver (include/mghip.h, "Cycle plans").

`DistributedMultigrid.cycle()` issues the same operations with the same pointers every cycle: fused legs, halo copies,
send/recv groups, the coarse gather, the replicated engine's cycle, the norm reduction.  The first cycle runs through the
Python driver with a `PlanRecorder` attached, which notes each operation as an `mg_plan_op`; every later cycle is one
`mg_plan_run` call that enqueues the whole list from C++ (kernels, RCCL calls, stream dependencies).  The reference has
no working counterpart (its DistributedMultigridSolver cannot be imported); what this replaces is the
interpreter time of our own per-cycle Python.
"""
import ctypes as C
import os

from . import _lib


class PlanRecorder:
    """Collects operations while one cycle executes eagerly.  `stream` = 0 (compute) / 1 (communication) is the stream the
    operations being recorded run on; tensors whose pointers are recorded are kept alive in `keep`."""

    def __init__(self):
        self.ops = []
        self.keep = []
        self.stream = 0
        self.n_comm = 0
        self.split = None        # index of the first operation of the BACK part of the cycle (mark_split)

    def mark_split(self):
        """Everything recorded from here on is the back part of the cycle: the level-0 up legs and the norm.  The front part
        (the level-0 down legs and everything below them) never writes the buffers that hold the iterate, so it can be queued
        for the NEXT cycle before this cycle's norm has reached the host and simply be left unused if the solve ends there."""
        if self.split is None:
            self.split = len(self.ops)

    def emit(self, op, i=(), d=(), p=(), keep=()):
        o = _lib.MgPlanOp()
        o.op, o.stream = int(op), int(self.stream)
        for k, v in enumerate(i):
            o.i[k] = int(v)
        for k, v in enumerate(d):
            o.d[k] = float(v)
        for k, v in enumerate(p):
            o.p[k] = v if (v is None or isinstance(v, int)) else v.data_ptr()
            if v is not None and not isinstance(v, int):
                self.keep.append(v)
        self.keep.extend(keep)
        self.ops.append(o)
        return o

    # ---- data movement -----------------------------------------------------------------------
    def copy2d(self, dst, src):
        if dst.dim() != 2 or src.shape != dst.shape or src.dtype != dst.dtype or \
                (dst.shape[1] > 1 and (dst.stride(1) != 1 or src.stride(1) != 1)):
            raise ValueError("plan copies are 2-D, unit-stride rows, same shape and dtype")
        es = dst.element_size()
        self.emit(_lib.MG_PLAN_COPY2D, i=(dst.shape[0], dst.shape[1] * es, dst.stride(0) * es, src.stride(0) * es), p=(dst, src))

    def add(self, dst, a, b=None):
        self.emit(_lib.MG_PLAN_ADD_F64, p=(dst, a, b))

    def group(self, sends, recvs):
        """one ncclGroup of sends and receives; (peer, contiguous tensor) pairs"""
        if not (sends or recvs):
            return
        self.emit(_lib.MG_PLAN_GROUP_BEGIN)
        for peer, t in sends:
            self._sendrecv(_lib.MG_PLAN_SEND, peer, t)
        for peer, t in recvs:
            self._sendrecv(_lib.MG_PLAN_RECV, peer, t)
        self.emit(_lib.MG_PLAN_GROUP_END)
        self.n_comm += 1

    def _sendrecv(self, op, peer, t):
        if not t.is_contiguous():
            raise ValueError("plan send/recv buffers are contiguous")
        self.emit(op, i=(peer, t.numel() * t.element_size()), p=(t,))

    def allgather(self, send, recv):
        if not (send.is_contiguous() and recv.is_contiguous()):
            raise ValueError("plan gather buffers are contiguous")
        self.emit(_lib.MG_PLAN_ALLGATHER, i=(send.numel() * send.element_size(),), p=(send, recv))
        self.n_comm += 1

    def allreduce(self, t):
        self.emit(_lib.MG_PLAN_ALLREDUCE_F64, i=(t.numel(),), p=(t,))
        self.n_comm += 1

    def result(self, t):
        self.emit(_lib.MG_PLAN_RESULT, p=(t,))

    # ---- stream dependencies -----------------------------------------------------------------
    def event_record(self, ev):
        self.emit(_lib.MG_PLAN_EVENT_RECORD, i=(ev,))

    def stream_wait(self, ev):
        self.emit(_lib.MG_PLAN_STREAM_WAIT, i=(ev,))

    def signature(self):
        """What a peer must mirror: per communication group, the (kind, peer, bytes) triples -- used by the tests to check
        that every send of one rank has its receive on the other, group by group."""
        groups, cur = [], None
        for o in self.ops:
            if o.op == _lib.MG_PLAN_GROUP_BEGIN:
                cur = []
            elif o.op in (_lib.MG_PLAN_SEND, _lib.MG_PLAN_RECV):
                cur.append(("send" if o.op == _lib.MG_PLAN_SEND else "recv", int(o.i[0]), int(o.i[1])))
            elif o.op == _lib.MG_PLAN_GROUP_END:
                groups.append(("p2p", tuple(cur)))
                cur = None
            elif o.op == _lib.MG_PLAN_ALLGATHER:
                groups.append(("allgather", int(o.i[0])))
            elif o.op == _lib.MG_PLAN_ALLREDUCE_F64:
                groups.append(("allreduce", int(o.i[0])))
        return groups


class RcclComm:
    """An RCCL communicator of the library's own (mg_comm_init), spanning the ranks of torch.distributed's default group.
    The unique id travels over torch.distributed; RCCL itself is the copy torch already loaded."""

    def __init__(self, dist, device_index):
        import torch
        self.lib = _lib.load()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.path = (path if os.path.exists(path) else "librccl.so.1").encode()
        world, rank = dist.get_world_size(), dist.get_rank()
        ident = (C.c_char * 128)()
        box = [None]
        if rank == 0:
            try:
                _lib.check_plan(self.lib.mg_comm_unique_id(self.path, ident))
                box = [bytes(ident.raw)]
            except Exception as exc:                 # every rank learns of it through the broadcast and raises too
                box = [exc]
        dist.broadcast_object_list(box, src=0)
        if not isinstance(box[0], bytes):
            raise RuntimeError(f"rank 0 could not create an RCCL unique id: {box[0]}")
        ident.raw = box[0]
        self.handle = C.c_void_p()
        _lib.check_plan(self.lib.mg_comm_init(self.path, ident, world, rank, int(device_index), C.byref(self.handle)))

    @classmethod
    def single(cls, device_index):
        """one-rank communicator (self-test of the binding on a one-GPU box)"""
        import torch
        self = cls.__new__(cls)
        self.lib = _lib.load()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.path = (path if os.path.exists(path) else "librccl.so.1").encode()
        ident = (C.c_char * 128)()
        _lib.check_plan(self.lib.mg_comm_unique_id(self.path, ident))
        self.handle = C.c_void_p()
        _lib.check_plan(self.lib.mg_comm_init(self.path, ident, 1, 0, int(device_index), C.byref(self.handle)))
        return self

    def ranks(self):
        """(ranks of the communicator, this process's rank in it)"""
        n, r = C.c_int(0), C.c_int(0)
        _lib.check_plan(self.lib.mg_comm_ranks(self.handle, C.byref(n), C.byref(r)))
        return n.value, r.value

    def close(self):
        if self.handle:
            self.lib.mg_comm_destroy(self.handle)
            self.handle = C.c_void_p()


_shared = {}
_shared_streams = {}


def shared_comm_stream(device):
    """ONE communication stream per process and device, used by every solver of that process (the fp32 and the fp64 one share
    the RCCL communicator of shared_comm, and every RCCL call on a communicator must come from one stream: with a stream per
    solver the fp32 solver's queued front part and the fp64 plan's groups after a precision switch would reach it from two)."""
    import torch
    key = torch.device(device).index or 0
    if key not in _shared_streams:
        _shared_streams[key] = torch.cuda.Stream(device=torch.device("cuda", key))
    return _shared_streams[key]


def shared_comm(dist, device_index):
    """one communicator per process and device, shared by the solvers of that process (the fp32 and the fp64 one)"""
    key = int(device_index)
    if key not in _shared:
        _shared[key] = RcclComm(dist, device_index)
    return _shared[key]


def shutdown():
    for c in _shared.values():
        c.close()
    _shared.clear()
    _shared_streams.clear()


class CyclePlan:
    """mg_plan built from a recorder; `run` enqueues it and returns the RESULT (None without one)."""

    def __init__(self, recorder, comm, device_index, lo=0, hi=None):
        """operations [lo, hi) of the recorder"""
        self.lib = _lib.load()
        self.keep = list(recorder.keep)
        ops = recorder.ops[lo:hi]
        self.n = len(ops)
        arr = (_lib.MgPlanOp * self.n)(*ops)
        self.handle = C.c_void_p()
        self.has_result = any(o.op == _lib.MG_PLAN_RESULT for o in ops)
        _lib.check_plan(self.lib.mg_plan_create(arr, self.n, comm.handle if comm is not None else None, int(device_index),
                                                C.byref(self.handle)))

    def run(self, compute_stream, comm_stream):
        out = C.c_double(0.0)
        rc = self.lib.mg_plan_run(self.handle, C.c_void_p(compute_stream), C.c_void_p(comm_stream), C.byref(out))
        if rc != _lib.MG_OK:
            _lib.check_plan(rc, self.handle)
        return out.value if self.has_result else None

    def run_async(self, compute_stream, comm_stream):
        """enqueue only; a RESULT is collected with wait()"""
        rc = self.lib.mg_plan_run_async(self.handle, C.c_void_p(compute_stream), C.c_void_p(comm_stream))
        if rc != _lib.MG_OK:
            _lib.check_plan(rc, self.handle)

    def wait(self):
        out = C.c_double(0.0)
        rc = self.lib.mg_plan_wait(self.handle, C.byref(out))
        if rc != _lib.MG_OK:
            _lib.check_plan(rc, self.handle)
        return out.value

    def profile(self, enable=True):
        """bracket every operation of the coming runs with timing events (diagnostics; see phase_times)"""
        _lib.check_plan(self.lib.mg_plan_profile(self.handle, int(bool(enable))), self.handle)

    def phase_times(self, into=None):
        """milliseconds per phase (_lib.PLAN_PHASE_NAMES) of the runs recorded since the last call, added to `into`"""
        out = (C.c_double * _lib.MG_PLAN_PHASES)()
        _lib.check_plan(self.lib.mg_plan_phase_times(self.handle, out), self.handle)
        res = into if into is not None else {}
        for k, name in enumerate(_lib.PLAN_PHASE_NAMES):
            res[name] = res.get(name, 0.0) + out[k]
        return res

    def copy_launches(self):
        """(COPY2D operations, launches they run as)"""
        a, b = C.c_int(0), C.c_int(0)
        _lib.check_plan(self.lib.mg_plan_copy_launches(self.handle, C.byref(a), C.byref(b)), self.handle)
        return a.value, b.value

    def close(self):
        if self.handle:
            self.lib.mg_plan_destroy(self.handle)
            self.handle = C.c_void_p()
        self.keep = []
