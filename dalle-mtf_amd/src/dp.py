"""Data-parallel gradient exchange: what Mesh-TensorFlow's lowering inserts for `layout: batch_dim:data`
(reference src/model_fns.py:81-82,189; the VAE's CrossShardOptimizer, src/model_fns_tf.py:61), made explicit.

One process per GPU.  The flat fp32 gradient buffer is reduced (SUM, in place) in buckets of at most 64 MB as soon as
their gradients are final, on a side HIP stream that an event orders after the bucket's last gradient kernel; clip + Adam
wait for the side stream.  Two transports:
  * "rccl"  -- RCCL over xGMI behind the C ABI (dmi_comm_init / dmi_allreduce_bucket, include/dalle_hip.h): the product path
               on a multi-GPU node;
  * "torch" -- torch.distributed.all_reduce(async_op=True) on a process group: the CPU (gloo) tests and ranks sharing one GPU
               in the GPU-box tests (DALLE_DP_TRANSPORT=torch).  It is NOT a silent fallback: when ranks own distinct GPUs a
               failing RCCL communicator is fatal unless DALLE_DP_STRICT=0.
xGMI is point to point (7 links x ~153 GB/s per GPU): a ring all-reduce of the 286 MB dalle_example gradient is ~3.3 ms
per-link bound against ~17 ms of compute per step, so the exchange hides behind backward as long as the last bucket is
small -- hence <= 64 MB pieces, issued in the order backward finishes them."""
from __future__ import annotations

import atexit
import os
from typing import List, Optional, Tuple

import torch

import dalle_hip as dh

MAX_BUCKET_BYTES = 64 << 20


def agree_ok(flag: bool, pg=None) -> bool:
    """True iff `flag` is true on EVERY rank (MIN all-reduce of a CPU scalar over the control-plane group); without a
    process group it is just `flag`.  Used wherever one rank's local failure must not leave the others in a collective."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return bool(flag)
    ok = torch.tensor([1 if flag else 0], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=pg)
    return int(ok.item()) == 1


def init_comm(world: int, rank: int, pg=None) -> Optional[int]:
    """Create this rank's RCCL communicator behind the C ABI.  The 128-byte unique id travels over the (CPU-capable)
    torch.distributed process group, which is control plane only.

    Failure policy: ranks that own distinct GPUs are the product path, and there an RCCL failure is FATAL on every rank
    (DALLE_DP_STRICT defaults to 1) -- a silent drop to torch.distributed would hide exactly the failure the first
    multi-GPU run has to surface.  DALLE_DP_STRICT=0 allows the collective fallback to the "torch" transport (returns None
    on every rank).  Order of events, so that no rank can be left alone in a blocking call:
      1. every rank binds librccl locally (dlopen + symbols), rank 0 also draws the unique id;     -> agree (MIN)
      2. the id is broadcast;  3. every rank enters dmi_comm_init (blocking collective);            -> agree (MIN)."""
    if world <= 1:
        return None
    import torch.distributed as dist
    strict = os.environ.get("DALLE_DP_STRICT", "1") != "0"

    def give_up(stage, err):
        msg = f"[dp] rank {rank}: RCCL {stage} did not succeed on every rank (this rank: {err or 'ok'})"
        if strict:
            raise RuntimeError(msg + "; set DALLE_DP_STRICT=0 to fall back to torch.distributed")
        print(msg + "; falling back to torch.distributed", flush=True)
        return None

    uid, err = None, None
    try:
        dh.comm_load()
        if rank == 0:
            uid = dh.comm_unique_id()
    except (dh.DalleHipError, OSError) as e:
        err = e
    if not agree_ok(err is None, pg):
        return give_up("library load", err)
    box = [uid]
    dist.broadcast_object_list(box, src=0, group=pg)
    handle = None
    try:
        handle = dh.comm_init(world, rank, box[0])
    except dh.DalleHipError as e:
        err = e
    if not agree_ok(handle is not None, pg):
        if handle:
            dh.comm_destroy(handle)
        return give_up("communicator init", err)
    atexit.register(_destroy_comm, handle)
    return handle


def _destroy_comm(handle):
    try:
        dh.comm_destroy(handle)
    except Exception:       # interpreter exit: the device context may already be gone
        pass


_SETUP = {}


def dist_setup():
    """(world, rank, process group, RCCL communicator handle or None) of this process; the communicator is created once.
    DALLE_DP_TRANSPORT=torch keeps the exchange on torch.distributed (tests with several ranks on one GPU)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0, None, None
    world, rank, pg = dist.get_world_size(), dist.get_rank(), dist.group.WORLD
    if "comm" not in _SETUP:
        want = os.environ.get("DALLE_DP_TRANSPORT", "rccl")
        _SETUP["comm"] = init_comm(world, rank, pg) if (want == "rccl" and torch.cuda.is_available()) else None
    return world, rank, pg, _SETUP["comm"]


def init_process_group(local_rank: int):
    """one process per GPU: gloo carries the control plane (object broadcasts, barriers, CPU scalars), nccl (= RCCL) is the
    torch-level fallback transport for device tensors; the gradient exchange itself uses the C-ABI communicator."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("DALLE_BENCH_SHARE_GPU") == "1":   # tests: several ranks on one GPU -> no RCCL at all
        os.environ["DALLE_DP_TRANSPORT"] = "torch"
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("cpu:gloo,cuda:nccl", device_id=torch.device("cuda", local_rank))
    return dist.group.WORLD


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


class GradReducer:
    """Bucketed all-reduce of one flat fp32 buffer.  ready(lo, hi) is called from backward, in completion order, as soon
    as g[lo:hi] is final on the CURRENT stream; finish() must precede the first consumer of the reduced gradients."""

    def __init__(self, flat_g: torch.Tensor, world: int, comm: Optional[int] = None, pg=None,
                 max_bucket_bytes: int = MAX_BUCKET_BYTES, timeout_s: Optional[float] = None):
        self.g, self.world, self.comm, self.pg = flat_g, world, comm, pg
        # [r06] A rank that fails inside ready() (a collective that raises, an error in the kernels before it) leaves its peers with
        # collectives it never joins.  They must not wait for ever: finish() waits at most `timeout_s` per collective on the torch
        # transport (DALLE_DP_TIMEOUT_S, default 600) and raises; a failing rank remembers its error, stops issuing and re-raises from
        # finish() as well, so that every rank leaves the step with an exception inside a bounded time (tests/test_dp_gloo.py).
        self.timeout_s = float(os.environ.get("DALLE_DP_TIMEOUT_S", "600")) if timeout_s is None else float(timeout_s)
        self.failed: Optional[BaseException] = None
        self.transport = "rccl" if comm else "torch"
        self.max_elems = max(1, max_bucket_bytes // 4)
        self.stream = torch.cuda.Stream(device=flat_g.device) if (comm and flat_g.is_cuda) else None
        self._pending: List = []          # torch transport: async work handles
        self._issued = False              # rccl transport: something is in flight on the side stream
        self.log: List[Tuple[int, int]] = []   # (lo, hi) of every collective issued since the last finish()
        self.last_log: List[Tuple[int, int]] = []   # the schedule of the previous step (tests, bench.py)

    def pieces(self, lo: int, hi: int) -> List[Tuple[int, int]]:
        out = []
        while lo < hi:
            nxt = min(hi, lo + self.max_elems)
            out.append((lo, nxt))
            lo = nxt
        return out

    def ready(self, lo: int, hi: int):
        if self.world <= 1 or hi <= lo:
            return
        if self.failed is not None:        # a previous piece of this step failed here: issue nothing more, finish() reports it
            return
        try:
            self._issue(lo, hi)
        except BaseException as e:
            self.failed = e
            raise

    def _issue(self, lo: int, hi: int):
        if self.transport == "rccl":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            for a, b in self.pieces(lo, hi):
                dh.allreduce_bucket(self.comm, self.g[a:b], b - a, stream=self.stream.cuda_stream)
                self.log.append((a, b))
            self._issued = True
        else:
            import torch.distributed as dist
            for a, b in self.pieces(lo, hi):
                self._pending.append(dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                self.log.append((a, b))

    def finish(self):
        """the current stream (clip + Adam) waits for every collective issued so far"""
        if self.transport == "rccl":
            if self._issued:
                ev = torch.cuda.Event()
                ev.record(self.stream)
                torch.cuda.current_stream().wait_event(ev)
                self._issued = False
        else:
            import datetime
            pending, self._pending = self._pending, []
            err = self.failed
            for i, h in enumerate(pending):
                if err is not None:
                    break                   # (this rank failed, or a peer did: the remaining handles can never complete)
                try:
                    h.wait(timeout=datetime.timedelta(seconds=self.timeout_s))
                except BaseException as e:
                    err = RuntimeError(f"[dp] collective {i + 1} of {len(pending)} of this step did not complete within {self.timeout_s:.0f} s "
                                       f"or failed -- a peer rank has left the gradient exchange: {e}")
            if err is not None:
                self.last_log, self.log, self.failed = self.log, [], None
                raise err
        if self.failed is not None:
            err, self.failed = self.failed, None
            self.last_log, self.log = self.log, []
            raise err
        self.last_log, self.log = self.log, []

    def broadcast(self, buf: torch.Tensor, root: int = 0):
        """initial weights / restored state from `root` to every rank"""
        if self.world <= 1:
            return
        if self.transport == "rccl":
            dh.comm_broadcast_f32(self.comm, buf, buf.numel(), root)
        else:
            import torch.distributed as dist
            dist.broadcast(buf, src=root, group=self.pg)
