"""Multi-GPU hypothesis sharding: one process per GPU, one exchange per frame.

The hypothesis list 0..H-1 is cut into ``world`` contiguous slices.  Every rank
holds the replicated frame (x, P, features, z), scores its slice (K2-K4), the
int32 supports are all-gathered (RCCL over xGMI when the tensors live in HBM,
gloo in the CPU tests), and every rank replays the sequential consensus scan
(K5) on the full list -- bit-identical on all ranks, so the reference's "earliest
strict maximum + adaptive stop" semantics (Tracking.cpp:403,507-537) survive the
sharding -- and applies the updates redundantly (no 26 MB covariance broadcast).
The only collective is that one all-gather of H * 4 bytes.

``mode="allreduce"`` (frames without the adaptive stop, adaptive = 0) replaces it by ONE 8-byte MAX all-reduce: a rank
folds its slice into ``key = support << 32 | (0xFFFFFFFF - index)``, the maximum over the ranks is the largest support
and, among equal supports, the smallest index -- the same earliest strict maximum -- and every rank replays the
consensus on the one-hot list {winner: its support} (the winner's inlier mask is recomputed locally).  The C ABI has the
same pair: rslam_shard_frame / rslam_shard_frame_allreduce.
"""
from typing import Protocol, Tuple

import torch
import torch.distributed as dist


def slice_bounds(H: int, rank: int, world: int) -> Tuple[int, int, int]:
    """(begin, end, chunk) of the contiguous slice of rank; chunk = ceil(H / world)."""
    chunk = (H + world - 1) // world
    b = min(H, rank * chunk)
    e = min(H, b + chunk)
    return b, e, chunk


class Engine(Protocol):
    """What the driver needs from a per-rank engine (HIP product or a test double)."""
    H: int
    device: torch.device

    def step_predict(self) -> None: ...
    def step_score(self, hyp_begin: int, hyp_end: int, local: torch.Tensor) -> None:
        """write supports of hypotheses [begin, end) into local[0 : end-begin] (int32)"""
    def step_update(self, supports_all: torch.Tensor) -> None: ...


KEY_INDEX_MASK = 0xFFFFFFFF


def slice_key(local: torch.Tensor, begin: int, n: int) -> torch.Tensor:
    """int64[1]: support << 32 | (0xFFFFFFFF - global index) of the slice's earliest maximum (0 for an empty slice);
    tensor ops only, so that on a GPU it stays on the stream (no host read-back)."""
    if n <= 0:
        return torch.zeros(1, dtype=torch.int64, device=local.device)
    sup = local[:n].to(torch.int64)
    idx = torch.arange(begin, begin + n, dtype=torch.int64, device=local.device)
    return ((sup << 32) | (KEY_INDEX_MASK - idx)).max().reshape(1)


def expand_key(key: torch.Tensor, out: torch.Tensor, H: int) -> None:
    """the one-hot support list of the reduced key: out[winner] = its support, every other entry of out[:H] zero"""
    out.zero_()
    if H > 0:
        h = (KEY_INDEX_MASK - (key & KEY_INDEX_MASK)).clamp_(0, H - 1)
        out.index_put_((h,), (key >> 32).to(out.dtype))


class ShardedFrame:
    def __init__(self, engine: Engine, group=None, mode: str = "allgather"):
        if mode not in ("allgather", "allreduce"):
            raise ValueError("mode is 'allgather' or 'allreduce'")
        self.engine = engine
        self.group = group
        self.mode = mode
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.resize()

    def resize(self):
        H = self.engine.H
        self.begin, self.end, self.chunk = slice_bounds(H, self.rank, self.world)
        dev = self.engine.device
        self.local = torch.zeros(max(self.chunk, 1), dtype=torch.int32, device=dev)
        self.all = torch.zeros(max(self.chunk, 1) * self.world, dtype=torch.int32, device=dev)

    def step(self):
        """One frame: predict, score own slice, exchange, consensus + updates."""
        ctx = getattr(self.engine, "stream_context", None)
        if ctx is None:
            return self._step()
        with ctx():                               # kernels and the collective on ONE stream: ordered without host syncs
            return self._step()

    def _step(self):
        e = self.engine
        if hasattr(e, "step_phase"):              # engines that replay each half from a hipGraph
            e.step_phase(0, self.begin, self.end, self.local)
        else:
            e.step_predict()
            e.step_score(self.begin, self.end, self.local)
        full = self.local
        if self.mode == "allreduce":
            # (adaptive = 0 only: the caller's engine evaluates every hypothesis)
            key = slice_key(self.local, self.begin, self.end - self.begin)
            if self.world > 1:
                dist.all_reduce(key, op=dist.ReduceOp.MAX, group=self.group)
            expand_key(key, self.all, e.H)
            full = self.all
        elif self.world > 1:
            dist.all_gather_into_tensor(self.all, self.local, group=self.group)
            full = self.all
        if hasattr(e, "step_phase"):
            e.step_phase(1, self.begin, self.end, full)
        else:
            e.step_update(full)


class HipEngine:
    """The product engine: ransac_slam_amd.api.RslamHip on this rank's GPU.  Kernels and the RCCL
    all-gather are enqueued on ONE torch stream, which orders them without host synchronisation.
    torch's default stream has the handle 0, which rslam_set_stream reads as "use the context's own
    stream" -- kernels would then run unordered with the collective -- so the engine never uses it:
    unless the caller's current stream is a real side stream it creates and owns one, and
    ShardedFrame.step() runs under it (stream_context)."""

    def __init__(self, ctx, device_index: int, use_graph: bool = True, stream=None):
        self.ctx = ctx
        self.use_graph = use_graph
        self.device = torch.device("cuda", device_index)
        if stream is None:
            cur = torch.cuda.current_stream(self.device)
            stream = cur if cur.cuda_stream != 0 else torch.cuda.Stream(device=self.device)
        if stream.cuda_stream == 0:
            raise ValueError("HipEngine needs a non-default torch stream")
        self.stream = stream
        ctx.set_stream(stream.cuda_stream)

    def stream_context(self):
        return torch.cuda.stream(self.stream)

    @property
    def H(self):
        return self.ctx.H

    def step_predict(self):
        self.ctx.step_predict()

    def step_score(self, hyp_begin, hyp_end, local):
        # the C ABI indexes the support array by global hypothesis id
        self.ctx.step_score(hyp_begin, hyp_end, local.data_ptr() - 4 * hyp_begin)

    def step_update(self, supports_all):
        self.ctx.step_update(supports_all.data_ptr())

    def step_phase(self, phase, hyp_begin, hyp_end, tensor):
        # phase 0 writes tensor[0 : end-begin] (the C ABI indexes by global hypothesis id), phase 1 reads the full list
        ptr = tensor.data_ptr() - 4 * hyp_begin if phase == 0 else tensor.data_ptr()
        self.ctx.step_phase(phase, hyp_begin, hyp_end, ptr, self.use_graph)
