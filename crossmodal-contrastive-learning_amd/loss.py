"""MI355X-native drop-in for the reference criterion.

Mirrors `trainer/loss.py:44-114` of amazon-science/crossmodal-contrastive-learning:

    criterion = CrossCLR_onlyIntraModality(temperature=0.03, negative_weight=0.8, logger=None)
    loss = criterion(video_features, text_features)      # [B, D], [B, D] -> 0-dim float64

Same constructor and forward signature, same `state_dict()` (one key, `logit_scale`), same
registered child (`criterion`), same public attributes read at call time (`temperature`,
`negative_w`, `logger`), float64 0-dim result on the inputs' device, gradients in the input dtype,
`RuntimeError` for mismatched batch sizes and non-2-D inputs.  Underneath, the ~25 eager ops and
the three host->device mask copies per step are replaced by five kernel launches through the
C-ABI in include/crossclr.h; no B x B tensor is ever materialised.  Like the reference's eager ops the
criterion is twice differentiable: under `create_graph=True` the HIP backward is recorded as a node whose
own backward is crossclr_second_order (Hessian-vector product in closed form on the device, exact fp32).

Keyword-only additions (defaults reproduce the reference's single-process behaviour):
  compute_mode   "auto" | "fp32" | "bf16".  fp32 = exact-fp32 MFMA; bf16 = bf16 operands with
                 fp32 accumulation (the BASELINE headline mode).  auto = bf16 when the global
                 batch is >= 1024 rows (where its error is ~1e-5 on the loss; a one-time warning says so
                 for fp32/fp64 inputs), fp32 below that and whenever max(1,|w|)/temperature > 128
                 (small temperatures: the two-pass soft-max regime, where bf16 cosines are too coarse).
  process_group  a torch.distributed group: the batch is the concatenation of every rank's rows
                 (equal count per rank); the returned loss is the GLOBAL loss on every rank and the
                 gradients are exactly d(global loss)/d(local rows).
  negative_scale (functional form) per-sample multipliers (k_video[b], k_text[b]) >= 0 of the sample's
                 exponential wherever it is an intra-modal NEGATIVE column (0 prunes it from the negative
                 set), and
  loss_weight    per-sample weights (w_video[b], w_text[b]) of the sample's own loss term (all ones = the
                 reference's mean).  Both are constants (no gradient); None = the reference's loss.  This
                 is the hook for influential-sample pruning / weighting (`influence.py`),
                 which the reference @ v1 does not contain.
"""
from __future__ import annotations

import ctypes
import os
import warnings
from typing import Optional

import torch
from torch import nn

from . import _native as nat

_IN_DTYPE = {torch.float32: nat.IN_F32, torch.float16: nat.IN_F16, torch.bfloat16: nat.IN_BF16,
             torch.float64: nat.IN_F64}
AUTO_BF16_MIN_GLOBAL_BATCH = 1024


def _ptr(t: Optional[torch.Tensor]):
    """Device address as a plain int (ctypes converts it to void* itself; building a c_void_p per argument cost ~10 us per step)."""
    return 0 if t is None else t.data_ptr()


_raw_current_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_for(t: torch.Tensor):
    """The current stream of the tensor's device as the raw handle the C-ABI takes (torch.cuda.current_stream() builds a Stream object per
    call: 12 us, twice per step -- tools/host_profile.py)."""
    if t.is_cuda:
        if _raw_current_stream is not None:
            return _raw_current_stream(t.get_device())
        return torch.cuda.current_stream(t.device).cuda_stream
    return 0


def _row_major(t: torch.Tensor) -> torch.Tensor:
    return t if t.stride(1) == 1 and t.stride(0) >= t.shape[1] else t.contiguous()


_warned_auto_bf16 = False


def _resolve_mode(compute_mode: str, global_batch: int, in_dtype=None, small_temperature: bool = False) -> int:
    global _warned_auto_bf16
    if compute_mode == "fp32":
        return nat.MODE_FP32
    if compute_mode == "bf16":
        return nat.MODE_BF16
    if compute_mode == "auto":
        # bf16 rounds a cosine to 2^-9; the logit carries that times 1/tau: at small temperatures (the two-pass regime) only
        # exact-fp32 products keep the 1e-3 loss bar, and the generic kernels run there in either mode anyway
        if global_batch < AUTO_BF16_MIN_GLOBAL_BATCH or small_temperature:
            return nat.MODE_FP32
        if in_dtype in (torch.float32, torch.float64) and not _warned_auto_bf16:
            _warned_auto_bf16 = True
            import warnings
            warnings.warn("CrossCLR compute_mode='auto': global batch >= %d, using bf16 operands with fp32 accumulation for the "
                          "similarity products of these %s inputs (loss within ~1e-5, gradients within ~5e-3 of the fp32 "
                          "reference); pass compute_mode='fp32' for exact-fp32 products. This message is shown once."
                          % (AUTO_BF16_MIN_GLOBAL_BATCH, str(in_dtype).replace("torch.", "")), stacklevel=3)
        return nat.MODE_BF16
    raise ValueError(f"compute_mode must be 'auto', 'fp32' or 'bf16', got {compute_mode!r}")


class _Workspace:
    """Everything one forward produces and the backward consumes (all caller-owned torch tensors)."""
    __slots__ = ("plan", "xhat", "xcols", "inv_norm", "diag", "logz", "rz", "wrz", "rz_cols", "wrz_cols",
                 "loss_sum", "temperature", "negative_w", "world", "rank", "in_dtype", "sharded",
                 "k_rows", "k_cols", "lw", "stats_work", "stash", "shift", "shift_cols", "prenormalized",
                 "saved_blocks", "recompute_ranges", "exchange", "k_work", "group", "partner_peers", "xf", "xf_all", "step", "lazy")

    def __getattr__(self, name):
        # (reached only for a slot that holds nothing yet.)  The step path leaves the views of its workspace -- which only tools, tests and the
        # fused projection's backward look at -- as (offset, bytes, dtype) entries and carves them on first use: eight dtype views per
        # forward cost the host ~40 us (tools/host_profile.py), as much as everything in front of the first launch.
        if name != "lazy":
            try:
                lazy = object.__getattribute__(self, "lazy")
            except AttributeError:
                lazy = None
            if lazy and name in lazy:
                off, nbytes, dt = lazy[name]
                lay, persistent, transient = self.step
                if off < lay.persistent_bytes:
                    val = _carve(persistent, off, nbytes, dt)
                elif transient is not None:
                    val = _carve(transient, off - lay.persistent_bytes, nbytes, dt)
                else:
                    val = None      # (lived in the transient region, which has been given back)
                setattr(self, name, val)
                return val
        raise AttributeError(name)

    def release_transient(self) -> None:
        """Step path: give the transient workspace region (saved exponentials, fragment-major copy, partial sums: most of the bytes) back to
        the allocator.  Called once nothing the step will still launch reads it: right after crossclr_step_forward with
        CROSSCLR_STEP_EAGER, after the first backward otherwise (a later backward through the same graph recomputes).  Stream-ordered:
        the caching allocator reuses the block on the stream the kernels were enqueued on."""
        if self.step is None or self.step[2] is None:
            return
        lay, persistent, _ = self.step
        self.step = (lay, persistent, None)
        lazy = self.lazy or {}
        for name, (off, _, _) in lazy.items():
            if off >= lay.persistent_bytes:
                setattr(self, name, None)


_plan_cache: dict = {}
_checked_shapes: dict = {}


def _plan_for(b: int, D: int, world: int, rank: int, mode: int):
    """crossclr_make_plan is pure in its arguments (and in tuning variables the library reads once): cache it."""
    if nat.injected_for_testing():
        return nat.make_plan(b, D, world, rank, mode)     # the CPU tests flip tuning variables between calls: their build re-reads them
    key = (b, D, world, rank, mode, nat.library_path())
    plan = _plan_cache.get(key)
    if plan is None:
        if len(_plan_cache) > 256:
            _plan_cache.clear()
        plan = _plan_cache[key] = nat.make_plan(b, D, world, rank, mode)
    return plan


def _check_equal_rows_per_rank(b: int, D: int, group, dev, stash_bytes: int = 0) -> bool:
    """Every rank must bring the same [b, D] (the gathered operand is world x one rank's packed operand).  A mismatch would
    otherwise surface as a hang or an opaque error inside all_gather_into_tensor; checked once per (group, shape).

    The same all-reduce settles whether EVERY rank can hold `stash_bytes` of saved exponentials (a trial allocation): the
    partner-gradient scheme ships blocks' transposed contributions instead of letting the partner recompute them, so a rank that could
    not allocate its stash would raise while its peers sit in a collective.  Returns that agreement (rank-invariant)."""
    import torch.distributed as dist
    key = (id(group), b, D, stash_bytes)
    got = _checked_shapes.get(key)
    if got is not None:
        return got
    ok = 1
    if stash_bytes > 0:
        try:
            trial = torch.empty(stash_bytes, dtype=torch.uint8, device=dev)
            del trial
        except torch.OutOfMemoryError:
            ok = 0
    t = torch.tensor([b, -b, D, -D, -ok], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    lo_b, hi_b, lo_d, hi_d = -int(t[1]), int(t[0]), -int(t[3]), int(t[2])
    if lo_b != hi_b or lo_d != hi_d:
        raise RuntimeError(f"CrossCLR (sharded): every rank must pass the same number of rows and columns; this rank has "
                           f"[{b}, {D}] but the group spans [{lo_b}..{hi_b}, {lo_d}..{hi_d}]")
    if len(_checked_shapes) > 64:
        _checked_shapes.clear()
    all_ok = _checked_shapes[key] = int(t[4]) == -1
    return all_ok


_last_exchange_mode = None    # what the most recent sharded forward used ("allgather" | "p2p" | "p2p_each"): read by bench.py
_comm_trace = None     # bench.py's diagnostic steps set this to a list: (tag, start event, end event) around every wait on a collective


def _traced_wait(tag: str, works) -> None:
    """`work.wait()` makes the compute stream wait for the collective; with tracing on, HIP events around it measure how long the
    stream actually sat idle (= communication NOT hidden behind compute)."""
    if not works:
        return
    tr = _comm_trace
    if tr is not None and torch.cuda.is_available():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for w in works:
            w.wait()
        e1.record()
        tr.append((tag, e0, e1))
    else:
        for w in works:
            w.wait()


class _OperandExchange:
    """How the ranks' packed operands reach each other (CROSSCLR_EXCHANGE; `mode` below).

    allgather (default)  one `all_gather_into_tensor`; forward and backward wait for the same collective.
    p2p                  (pairs scheme only) the forward of rank r touches only the ranks it evaluates itself -- r+1 .. r+K and the
                         antipodal rank -- so those slices travel first (ONE batch of isend / irecv: every xGMI link busy at once);
                         the slices only the backward's recompute needs (r-K .. r-1) follow in a second batch that has the whole
                         remote forward to hide behind.  Same bytes, less of them on the forward's critical path (4/7 at 8 ranks).
    p2p_each             one isend / irecv pair PER PEER DISTANCE, forward-critical distances first, each with its own completion:
                         `wait_peer(r)` returns as soon as rank r's slice has landed, so the block against r can start while later
                         slices are still on the wire ("remote column tiles start as each peer's slice lands").
                         The price: a communicator executes its operations in order, so the distances go out one after another
                         (one link at a time) instead of all links at once.
    Which one wins on a real 8-GPU node is an open measurement: `bench.py --gpus N` reports the exchange it used and the time the
    compute stream spent waiting for it."""

    def __init__(self, xcols, xhat, group, world, rank, first_peers=None, each=False, defer_late=False):
        """defer_late: the slices no forward block needs (r-K .. r-1) are only requested if somebody asks for them (`wait()`): with
        partner gradients (crossclr_backward_rect_saved_t) nobody does, and half of the operand traffic never happens.  Every rank
        must take the same decision (it is derived from the plan and the environment only)."""
        import torch.distributed as dist
        self.mode = "allgather" if first_peers is None else ("p2p_each" if each else "p2p")
        self._first, self._rest, self._peer = [], [], {}
        self._deferred = None
        if first_peers is None:
            self._first = [dist.all_gather_into_tensor(xcols, xhat, group=group, async_op=True)]
            return
        n = xhat.numel()
        piece = lambda r: xcols[r * n:(r + 1) * n]
        piece(rank).copy_(xhat)
        # gloo has no device path for send / recv (its collectives stage device tensors through pinned memory, its point-to-point operations
        # hand the raw pointer to the socket layer: the HOST then writes device memory behind the GPU's L2, and a block launched right after
        # can read lines the L2 still holds from the buffer's previous life -- seen once in ~40 runs of tests/test_gpu_ranks_share_gpu.py).
        # Ranks that share one GPU over gloo (tests, `bench.py --share-gpu`) therefore stage these transfers themselves: pinned host
        # buffers on the wire, a stream-ordered copy into the gathered operand once a slice has landed.  RCCL and CPU tensors: untouched.
        self._staged = xhat.is_cuda and dist.get_backend(group) == "gloo"
        self._copies_first, self._copies_rest, self._copies_peer = [], [], {}
        if self._staged:
            xsend = xhat.cpu()
            def recv_target(r, copies):
                buf = torch.empty(n, dtype=xhat.dtype, pin_memory=True)
                copies.append((piece(r), buf))
                return buf
        else:
            xsend = xhat
            recv_target = lambda r, copies: piece(r)
        others = [(rank + d) % world for d in range(1, world)]
        early = [r for r in others if r in first_peers]
        late = [r for r in others if r not in first_peers]
        to_global = (lambda r: dist.get_global_rank(group, r)) if group is not None and group is not dist.group.WORLD else (lambda r: r)
        if each:
            # distance d: receive from rank + d, send to rank - d -- every rank posts the distances in the SAME order (the order in
            # which rank r needs rank r + d), so the pairs match up without a deadlock
            def post(peers):
                for r in peers:
                    d = (r - rank) % world
                    ops = [dist.P2POp(dist.irecv, recv_target(r, self._copies_peer.setdefault(r, [])), to_global(r), group),
                           dist.P2POp(dist.isend, xsend, to_global((rank - d) % world), group)]
                    self._peer[r] = dist.batch_isend_irecv(ops)
            post(early)
            if defer_late:
                self._deferred = lambda: post(late)
            else:
                post(late)
            return
        # rank s receives this rank's slice early iff this rank is one of ITS first peers: the offsets are symmetric
        offs_early = {(q - rank) % world for q in early}
        send_early = [(rank - d) % world for d in sorted(offs_early)]
        send_late = [r for r in others if r not in send_early]
        def post(recv_from, send_to, works, copies):
            ops = [dist.P2POp(dist.irecv, recv_target(r, copies), to_global(r), group) for r in recv_from]
            ops += [dist.P2POp(dist.isend, xsend, to_global(r), group) for r in send_to]
            if ops:
                works.extend(dist.batch_isend_irecv(ops))
        post(early, send_early, self._first, self._copies_first)
        if defer_late:
            self._deferred = lambda: post(late, send_late, self._rest, self._copies_rest)
        else:
            post(late, send_late, self._rest, self._copies_rest)

    @staticmethod
    def _land(copies):
        """(ranks sharing one GPU over gloo) the slices that have arrived in pinned host buffers go to their place, in stream order"""
        for dst, buf in copies:
            dst.copy_(buf, non_blocking=True)
        del copies[:]

    def wait_peer(self, r):
        """The slice of rank r has landed (p2p_each: exactly that; otherwise: the batch it travels in)."""
        if self.mode == "p2p_each":
            _traced_wait("operands:peer", self._peer.pop(r, None))
            self._land(self._copies_peer.pop(r, []))
        else:
            self.wait_forward()

    def wait_forward(self):
        _traced_wait("operands:forward", self._first)
        self._first = []
        if self.mode != "allgather":
            self._land(self._copies_first)

    def finish(self):
        """Everything that was REQUESTED has landed (deferred slices stay unrequested): buffers may be released."""
        self._deferred, deferred = None, self._deferred
        self.wait()
        self._deferred = None
        del deferred

    def wait(self):
        if self._deferred is not None:      # somebody needs the late slices after all (the recomputing backward): request them now
            self._deferred()
            self._deferred = None
        self.wait_forward()
        _traced_wait("operands:late", self._rest)
        self._rest = []
        if self.mode != "allgather":
            self._land(self._copies_rest)
        for r in list(self._peer):
            _traced_wait("operands:peer", self._peer.pop(r))
            self._land(self._copies_peer.pop(r, []))


# the pairs scheme's column-sum exchange: (device, process group, stream, world, rank, n2, npairs) -> (outbox, inbox, send / receive split
# sizes); the buffers are reused from step to step (a step's all-to-all has completed -- on the compute stream's order -- before the next
# step's kernels overwrite the outbox).  Keyed by group AND stream: two criteria on two streams, or on two process groups, never share them.
_pair_buffers: dict = {}


def _pair_exchange_buffers(dev, group, stream, world, rank, n2, npairs):
    key = (str(dev), id(group), int(stream), world, rank, n2, npairs)
    got = _pair_buffers.get(key)
    if got is None:
        if len(_pair_buffers) > 16:
            _pair_buffers.clear()
        # rank r evaluates the blocks against r+1 .. r+npairs and owes each of them n2 column sums; it is owed n2 sums by each of
        # r-npairs .. r-1.  all_to_all_single moves exactly those rows (split sizes: n2 for a partner, 0 otherwise); rows are
        # ordered by peer rank on both sides.
        send_to = sorted((rank + 1 + k) % world for k in range(npairs))
        recv_from = sorted((rank - 1 - k) % world for k in range(npairs))
        in_split = [n2 if r in send_to else 0 for r in range(world)]
        out_split = [n2 if r in recv_from else 0 for r in range(world)]
        outbox = torch.empty(npairs, n2, dtype=torch.float32, device=dev)
        inbox = torch.empty(npairs, n2, dtype=torch.float32, device=dev)
        got = _pair_buffers[key] = (outbox, inbox, in_split, out_split, {r: i for i, r in enumerate(send_to)})
    return got


# partner gradients: (outgoing [npairs, 2 bpad Dpad], incoming the same, one gradient-slice scratch) per (device, group, stream, shape),
# reused from step to step like the column-sum buffers (the step's all-to-all is waited for before its backward ends)
_partner_buffers: dict = {}


def _partner_gradient_buffers(dev, group, stream, npairs, nel, gbuf_floats):
    key = (str(dev), id(group), int(stream), npairs, nel, gbuf_floats)
    got = _partner_buffers.get(key)
    if got is None:
        if len(_partner_buffers) > 8:
            _partner_buffers.clear()
        got = _partner_buffers[key] = (torch.empty(npairs, nel, dtype=torch.float32, device=dev),
                                       torch.empty(npairs, nel, dtype=torch.float32, device=dev),
                                       torch.empty(gbuf_floats, dtype=torch.float32, device=dev))
    return got


# How the packed operands travel when nothing is said (CROSSCLR_EXCHANGE unset): from 3 ranks on -- where the pair scheme applies and a
# rank's forward touches only the K ranks after it and the antipode -- the need-ordered point-to-point form ("p2p": 4 of 7 slices on the
# forward's critical path at 8 ranks, every xGMI link busy at once; with partner gradients the other 3 are never requested), otherwise one
# all-gather.  `bench.py --gpus N` measures all three forms back to back on the node it runs on (set_exchange_for_benchmark).
_exchange_override = None


def set_exchange_for_benchmark(mode):
    """bench.py: force "allgather" / "p2p" / "p2p_each" for the following steps (None: back to the default rule)."""
    global _exchange_override
    if mode not in (None, "allgather", "p2p", "p2p_each"):
        raise ValueError(mode)
    _exchange_override = mode


def _exchange_mode(world: int, pairs_apply: bool) -> str:
    if _exchange_override is not None:
        return _exchange_override
    e = os.environ.get("CROSSCLR_EXCHANGE")
    if e in ("allgather", "p2p", "p2p_each"):
        return e
    return "p2p" if (world >= 3 and pairs_apply) else "allgather"


class _Range:
    """roctx range around one stage of the step (rocprofv3 --marker-trace shows them); only with CROSSCLR_ROCTX=1 -- the push / pop
    pair costs ~2 us of host time per stage.  `torch.cuda.nvtx` is roctx on ROCm builds of torch."""
    enabled = os.environ.get("CROSSCLR_ROCTX") == "1"

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        if _Range.enabled:
            torch.cuda.nvtx.range_push(self.name)

    def __exit__(self, *exc):
        if _Range.enabled:
            torch.cuda.nvtx.range_pop()


def _carve(total: torch.Tensor, offset: int, nbytes: int, dtype):
    return total[offset:offset + nbytes].view(dtype)


def _pack_pair(pair, b: int, bpad: int, dev, what: str) -> Optional[torch.Tensor]:
    """(video[b], text[b]) -> float32 [2][bpad] in the statistics layout, zero padded."""
    if pair is None:
        return None
    packed = getattr(pair, "packed", None)   # influence.PackedPair: already in the kernels' layout
    if packed is not None:
        if packed.numel() != 2 * bpad or packed.dtype != torch.float32 or packed.device != dev or pair.b != b:
            raise ValueError(f"{what}: packed weights do not match this batch")
        return packed
    if not (isinstance(pair, (tuple, list)) and len(pair) == 2):
        raise ValueError(f"{what} must be a (video[b], text[b]) pair of 1-D tensors")
    out = torch.zeros(2, bpad, dtype=torch.float32, device=dev)
    for m, x in enumerate(pair):
        if not torch.is_tensor(x) or x.dim() != 1 or x.shape[0] != b:
            raise ValueError(f"{what}[{m}] must be a 1-D tensor with one entry per local sample ({b})")
        out[m, :b] = x.detach().to(device=dev, dtype=torch.float32)
    return out.view(-1)


# Exact-fp32 plans advertise a stash of (2 bpad)^2 fp32 exponentials (1 GiB at b = 8192, 16 GiB at b = 32768; twice that in the
# two-pass regime).  Above this many bytes the step recomputes instead of saving (CROSSCLR_MAX_STASH_GB, default 8); and an
# allocation the device cannot satisfy falls back to the recomputing backward as well instead of failing the step.
_MAX_STASH_BYTES = int(float(os.environ.get("CROSSCLR_MAX_STASH_GB", "8")) * (1 << 30))


# Embedding widths (padded) at which the step takes the fragment-major saved backward (crossclr_backward_saved_xf) instead of the
# LDS-staged one.  The library offers it for every Dpad <= 1024 (plan.xf_bytes); measured on the MI355X (profiles/r03_xf_widths.txt) it
# wins from 512 up (D = 512, B = 2048 / 8192 / 16384: -6 / -3 ... -9 / -1 %; D = 768 / 1024 at B = 8192: -2 / -4 %), ties at 384 and
# loses at 256 / 128 (+14 / +21 % at B = 8192: with short tiles the saved exponentials' HBM latency is no longer covered -- the fragment
# loads must complete inside their iteration, which caps the exponentials' prefetch distance at about one tile).  The second copy costs
# crossclr_normalize_xf +1.5 us at D = 512 and +5 us at D = 1024 (+8 at B = 2048, where its 16-row blocks no longer fill the chip), so small
# batches keep the plain pair: default = Dpad in {512, 768, 1024} with at least 2048 (D <= 512) / 4096 (wider) padded rows.
# CROSSCLR_XF_WIDTHS="128,256,384,512" (or "" for none) overrides the widths and drops the row floor (tuning / tests).
_XF_WIDTHS_DEFAULT = frozenset((512, 768, 1024, 1152, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 8192))     # (beyond 1024: wide plans, pair kernel only)


# The fragment-major backwards move their column tiles with hand-counted inline-asm loads, and two schedules of that idea that compiled
# and audited clean were wrong on the hardware.  The committed ones are soaked per instantiation (tools/soak_xf.py) -- with
# THIS compiler; a compiler upgrade means re-running that soak.  The module does not take a library's word for it either: before a
# process takes one of them at a kernel instantiation (device, padded width, sample weights, library) it runs a SELF-TEST outside
# any training step -- synthetic batches of 640 and 1152 rows (mirrored and direct tiles, both loop phases, an odd and an even number
# of 256-row blocks), forward + finish, then the candidate against the LDS-staged kernel over the same saved exponentials, several launches each,
# gradient buffers compared bit for bit (a few milliseconds, once).  A difference disables the candidate for the process with a warning
# (crossclr_backward_saved_xfp falls back to crossclr_backward_saved_xf, that one to the LDS-staged kernel).  Under HIP-graph capture
# nothing can be verified: an unverified instantiation then takes the LDS-staged kernel.  CROSSCLR_XF_RECHECK_STEPS=N repeats the
# self-test every N backward launches of an instantiation (default 0: never).  The tests' injected build (host emulation: no such
# hazard exists there) skips the check.
_xf_verified: dict = {}
_xf_launches: dict = {}
_XF_RECHECK = int(os.environ.get("CROSSCLR_XF_RECHECK_STEPS", "0") or 0)
_XF_SELFTEST_ROWS = (640, 1152)


def _xf_selftest(dev, D: int, weighted: bool, entry_name: str, launches: int = 3) -> bool:
    """True when `entry_name` (a fragment-major saved backward) reproduces crossclr_backward_saved bit for bit on this device."""
    lib = nat.library()
    cand = getattr(lib, entry_name)
    stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
    ok = True
    for b in _XF_SELFTEST_ROWS:
        plan = nat.make_plan(b, D, 1, 0, nat.MODE_BF16)
        if not (plan.stash_bytes > 0 and plan.xf_bytes > 0):
            return None        # nothing to compare on this build / under these knobs: not a mismatch (the caller keeps the LDS-staged kernel, silently)
        pp = ctypes.byref(plan)
        g = torch.Generator().manual_seed(1000 + b)
        # aligned pairs (t = v + noise): a soft-max that is neither flat nor one-hot, so every tile carries weights of mixed magnitude
        v = torch.randn(b, D, generator=g)
        t = (v + 0.5 * torch.randn(b, D, generator=g)).to(dev)
        v = v.to(dev)
        n2 = 2 * plan.bpad
        f32 = dict(dtype=torch.float32, device=dev)
        xhat = torch.empty(plan.operand_bytes, dtype=torch.uint8, device=dev)
        xf = torch.empty(plan.xf_bytes, dtype=torch.uint8, device=dev)
        inv_norm, diag = torch.empty(n2, **f32), torch.empty(plan.bpad, **f32)
        logz, rz, wrz = torch.empty(n2, **f32), torch.empty(n2, **f32), torch.empty(n2, **f32)
        part = torch.empty(plan.fwd_ws_floats, **f32)
        loss_sum = torch.empty(max(2, plan.loss_ws_doubles), dtype=torch.float64, device=dev)
        stash = torch.empty(plan.stash_bytes, dtype=torch.uint8, device=dev)
        k = None
        if weighted:
            k = torch.zeros(2, plan.bpad, **f32)
            k[:, :b] = (torch.rand(2, b, generator=g) > 0.3).float().to(dev)
            k = k.view(-1)
        sw = _sw(k, k, None)
        nat.check(lib.crossclr_normalize_xf(pp, _ptr(v), _ptr(t), v.stride(0), t.stride(0), nat.IN_F32, _ptr(xhat), _ptr(xf), _ptr(inv_norm),
                                            _ptr(diag), stream))
        nat.check(lib.crossclr_forward_save(pp, _ptr(xhat), 0.05, 0.8, sw, _ptr(part), 0, _ptr(stash), stream))
        nat.check(lib.crossclr_forward_finish_w(pp, _ptr(part), plan.fwd_slots, _ptr(diag), 0.05, 0.8, sw, _ptr(logz), _ptr(rz), _ptr(wrz),
                                                _ptr(loss_sum), stream))
        want = torch.empty(plan.gbuf_bytes // 4, **f32)
        nat.check(lib.crossclr_backward_saved(pp, _ptr(xhat), _ptr(stash), 0.05, 0.8, _ptr(rz), _ptr(wrz), sw, _ptr(want), 0, stream))
        for _ in range(launches):
            got = torch.full_like(want, float("nan"))
            nat.check(cand(pp, _ptr(xf), _ptr(stash), 0.05, 0.8, _ptr(rz), _ptr(wrz), sw, _ptr(got), 0, stream))
            ok = ok and bool(torch.equal(got, want))
    return ok


def _saved_backward_entry(ws, plan, dev) -> str:
    """Which saved backward of the local block this step launches: "crossclr_backward_saved_xfp" (pairs of tiles per barrier),
    "crossclr_backward_saved_xf", or "crossclr_backward_saved" (column tiles staged through LDS; reads the row-major operand)."""
    if ws.xf is None:
        return "crossclr_backward_saved"
    weighted = ws.k_rows is not None
    cands = ["crossclr_backward_saved_xf"] if plan.Dpad <= 1024 else []     # (wide plans: the pair kernel or the LDS-staged one)
    if plan.stash_bytes < (1 << 32) and os.environ.get("CROSSCLR_XFP", "1") != "0":
        cands.insert(0, "crossclr_backward_saved_xfp")
    for name in cands:
        key = (str(dev), plan.Dpad, weighted, name, nat.library_path())
        ok = _xf_verified.get(key)
        if ok and _XF_RECHECK > 0:
            n = _xf_launches[key] = _xf_launches.get(key, 0) + 1
            if n % _XF_RECHECK == 0 and not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
                ok = None
        if ok is None:
            if nat.injected_for_testing():
                ok = True
            elif dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
                return "crossclr_backward_saved"          # nothing can be verified inside a capture
            else:
                ok = _xf_selftest(dev, plan.D, weighted, name)
                if ok is False:
                    import warnings
                    warnings.warn(f"CrossCLR: {name} disagrees with the LDS-staged saved backward on this device / build at Dpad = {plan.Dpad}; "
                                  "it is disabled for this process (tools/soak_xf.py reproduces the comparison)")
            _xf_verified[key] = ok = bool(ok)
        if ok:
            return name
    return "crossclr_backward_saved"


def _use_xf(plan) -> bool:
    e = os.environ.get("CROSSCLR_XF_WIDTHS")
    if e is None:
        return plan.Dpad in _XF_WIDTHS_DEFAULT and plan.bpad >= (2048 if plan.Dpad <= 512 else 4096)
    return plan.Dpad in frozenset(int(x) for x in e.split(",") if x.strip())


def _alloc_stash(nbytes: int, dev) -> Optional[torch.Tensor]:
    if nbytes <= 0 or nbytes > _MAX_STASH_BYTES:
        return None
    try:
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)
    except torch.OutOfMemoryError:
        return None


def _alloc_rect_stash(nbytes: int, dev) -> torch.Tensor:
    """The saved exponentials of a block against other ranks.  The ranks agreed on the scheme from a trial allocation at the first step
    (`_check_equal_rows_per_rank`); memory pressure later cannot be renegotiated inside a step -- the peers already sit in their collectives --
    so a failure names the knobs instead of surfacing as a bare OOM on one rank and a hang on the others."""
    try:
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)
    except torch.OutOfMemoryError as e:
        raise RuntimeError(f"CrossCLR (sharded): could not allocate {nbytes / 2 ** 20:.0f} MiB for the saved exponentials of a remote block on this "
                           "rank (the other ranks are waiting in a collective and will time out); set CROSSCLR_DISABLE_REMOTE_SAVE=1 (remote "
                           "blocks recompute in the backward) or CROSSCLR_DISABLE_SAVE=1 on EVERY rank") from e


def _sw(k_rows, k_cols, lw) -> "ctypes.POINTER(nat.SampleWeights) | None":
    if k_rows is None and lw is None:
        return None
    s = nat.SampleWeights(0 if k_rows is None else k_rows.data_ptr(), 0 if k_cols is None else k_cols.data_ptr(),
                          0 if lw is None else lw.data_ptr())
    return ctypes.pointer(s)


_last_step_backward_kernel = None     # crossclr_step_layout.backward_kernel of the most recent single-device forward (tests, bench.py)
_last_step_saved = None               # ... and whether that step saved its exponentials


def _step_flags(plan, flags: int, temperature: float, negative_w: float, weighted: bool, dev) -> "tuple[int, nat.StepLayout]":
    """The single-device step's flags after this module's one reservation about the library's choice: a hand-scheduled fragment-major
    backward (crossclr_backward_saved_xfp / _xf) is only taken once it has reproduced the LDS-staged kernel bit for bit on this device
    (`_xf_selftest`, once per instantiation); a rejected -- or, inside a HIP-graph capture, unverifiable -- one is taken out with
    CROSSCLR_STEP_NO_XFP / _NO_XF and the library plans again.  Everything else (two-pass regime, save or recompute, layouts) is
    crossclr_step_plan's decision."""
    lib = nat.library()
    lay = nat.StepLayout()
    names = {3: ("crossclr_backward_saved_xfp", nat.STEP_NO_XFP), 2: ("crossclr_backward_saved_xf", nat.STEP_NO_XF)}
    while True:
        nat.check(lib.crossclr_step_plan(ctypes.byref(plan), temperature, negative_w, flags, 0, ctypes.byref(lay)))
        cand = names.get(lay.backward_kernel)
        if cand is None or nat.injected_for_testing():
            return flags, lay
        name, off_flag = cand
        key = (str(dev), plan.Dpad, weighted, name, nat.library_path())
        ok = _xf_verified.get(key)
        if ok and _XF_RECHECK > 0:
            n = _xf_launches[key] = _xf_launches.get(key, 0) + 1
            if n % _XF_RECHECK == 0 and not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
                ok = None
        if ok is None:
            if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
                flags |= nat.STEP_NO_XFP | nat.STEP_NO_XF       # nothing can be verified inside a capture: the LDS-staged kernel
                continue
            verdict = _xf_selftest(dev, plan.D, weighted, name)
            if verdict is False:
                warnings.warn(f"CrossCLR: {name} disagrees with the LDS-staged saved backward on this device / build at Dpad = {plan.Dpad}; "
                              "it is disabled for this process (tools/soak_xf.py reproduces the comparison)")
            ok = _xf_verified[key] = bool(verdict)
        if ok:
            return flags, lay
        flags |= off_flag


def _forward_step(video, text, temperature, negative_w, plan, mode, negative_scale, loss_weight, save_for_backward, prenormalized):
    """The single-device step through the library's two calls (include/crossclr.h, ABI 6): crossclr_step_forward here,
    crossclr_step_backward in `_backward_step`.  One caller-owned workspace whose layout the library reports; the module carves views of
    it for the tools and tests that look at intermediate results."""
    lib = nat.library()
    dev = video.device
    b = video.shape[0]
    ws = _Workspace()
    ws.plan, ws.world, ws.rank, ws.sharded = plan, 1, 0, False
    ws.temperature, ws.negative_w = float(temperature), float(negative_w)
    ws.in_dtype = _IN_DTYPE[video.dtype]
    ws.group = ws.partner_peers = ws.exchange = ws.k_work = ws.stats_work = None
    ws.saved_blocks = ws.recompute_ranges = ws.xf_all = None
    ws.k_rows = _pack_pair(negative_scale, b, plan.bpad, dev, "negative_scale")
    ws.lw = _pack_pair(loss_weight, b, plan.bpad, dev, "loss_weight")
    ws.k_cols = ws.k_rows
    ws.prenormalized = bool(prenormalized)
    flags = (nat.STEP_PRENORMALIZED if prenormalized else 0) | (0 if save_for_backward else nat.STEP_FORWARD_ONLY)
    # With a backward to follow the forward call also enqueues the gradient product (it does not depend on grad_out): the GPU keeps working
    # while the host walks from forward() to backward() -- autograd's thread hand-over -- and backward() is the finish kernel alone.
    # CROSSCLR_EAGER_BACKWARD=0: the product is launched from backward(), as in rounds 1-4.
    if save_for_backward and os.environ.get("CROSSCLR_EAGER_BACKWARD", "1") != "0":
        flags |= nat.STEP_EAGER
    flags, lay = _step_flags(plan, flags, ws.temperature, ws.negative_w, ws.k_rows is not None, dev)
    # Two caller-owned regions (include/crossclr.h, ABI 7): `persistent` is what crossclr_step_backward reads (with the eager gradient product:
    # 1 / ||x|| and the gradient slices), `transient` the rest -- saved exponentials, fragment-major copy, partial sums -- which the autograd
    # function gives back as soon as nothing reads it any more (`_Workspace.release_transient`).
    def regions():
        return (torch.empty(lay.persistent_bytes, dtype=torch.uint8, device=dev) if lay.persistent_bytes else None,
                torch.empty(lay.transient_bytes, dtype=torch.uint8, device=dev))
    try:
        persistent, transient = regions()
    except torch.OutOfMemoryError:
        if not lay.saved:
            raise
        persistent = transient = None
        flags |= nat.STEP_NO_SAVE          # the saved exponentials do not fit: the recomputing pair
        nat.check(lib.crossclr_step_plan(ctypes.byref(plan), ws.temperature, ws.negative_w, flags, 0, ctypes.byref(lay)))
        persistent, transient = regions()
    # (the loss the caller gets back is a 0-dim view of this buffer: its own small allocation, so it never pins the workspace)
    ws.loss_sum = torch.empty(max(2, plan.loss_ws_doubles), dtype=torch.float64, device=dev)
    with _Range("crossclr.step_forward"):
        nat.check(lib.crossclr_step_forward(ctypes.byref(plan), ctypes.byref(lay), _ptr(video), _ptr(text), video.stride(0), text.stride(0),
                                            ws.in_dtype, _sw(ws.k_rows, ws.k_rows, ws.lw), _ptr(persistent), _ptr(transient),
                                            _ptr(ws.loss_sum), _stream_for(video)))
    n2 = 2 * plan.bpad
    ws.step = (lay, persistent, transient)
    ws.lazy = lazy = {}
    for names, off, nbytes, dt in ((("xhat", "xcols"), lay.xhat, plan.operand_bytes, torch.uint8),
                                   (("inv_norm",), lay.inv_norm, 4 * n2, torch.float32), (("diag",), lay.diag, 4 * plan.bpad, torch.float32),
                                   (("logz",), lay.logz, 4 * n2, torch.float32), (("rz", "rz_cols"), lay.rz, 4 * n2, torch.float32),
                                   (("wrz", "wrz_cols"), lay.wrz, 4 * n2, torch.float32), (("shift", "shift_cols"), lay.shift, 4 * n2, torch.float32),
                                   (("xf",), lay.xf, lay.xf_bytes, torch.uint8), (("stash",), lay.stash, lay.stash_bytes, torch.uint8)):
        for name in names:
            if off == nat.STEP_NONE:
                setattr(ws, name, None)
            else:
                lazy[name] = (off, nbytes, dt)      # carved on first use (_Workspace.__getattr__)
    global _last_step_backward_kernel, _last_step_saved
    _last_step_backward_kernel, _last_step_saved = lay.backward_kernel, bool(lay.saved)
    return ws.loss_sum[1], ws      # = sum / (2 B), written by the finish kernel


def _backward_step(ws, video, text, grad_out):
    lib = nat.library()
    plan = ws.plan
    dev = video.device
    lay, persistent, transient = ws.step
    scratch = torch.empty(lay.backward_scratch_bytes, dtype=torch.uint8, device=dev) if lay.backward_scratch_bytes else None
    if grad_out.dtype == torch.float64 and grad_out.device == dev and grad_out.numel() == 1:
        go = grad_out.detach().reshape(1)
    else:
        go = grad_out.detach().to(device=dev, dtype=torch.float64).reshape(1).contiguous()
    gv = torch.empty(video.shape, dtype=video.dtype, device=dev)
    gt = torch.empty(text.shape, dtype=text.dtype, device=dev)
    with _Range("crossclr.step_backward"):
        nat.check(lib.crossclr_step_backward(ctypes.byref(plan), ctypes.byref(lay), _ptr(video), _ptr(text), video.stride(0), text.stride(0),
                                             ws.in_dtype, _sw(ws.k_rows, ws.k_rows, ws.lw), _ptr(persistent), _ptr(transient),
                                             _ptr(scratch), _ptr(go), _ptr(gv), _ptr(gt), gv.stride(0), gt.stride(0), _stream_for(video)))
    # (without the eager gradient product this launch was the last reader of the saved exponentials: a second backward through the same graph
    #  -- retain_graph=True -- finds transient == None and the library recomputes the product from the persistent region)
    ws.release_transient()
    return gv, gt


def _forward_impl(video: torch.Tensor, text: torch.Tensor, temperature: float, negative_w: float,
                  compute_mode: str, group, negative_scale=None, loss_weight=None,
                  save_for_backward: bool = False, prenormalized: bool = False, project=None) -> "tuple[torch.Tensor, _Workspace]":
    """project = (w_video_bf16, w_text_bf16 [fragment-major: projection._weights_bf16(w, Dpad)], (ldw_video, ldw_text), bias_video, bias_text, D): `video` / `text` are the projection head's INPUTS
    [b, Din]; the packed operand comes from crossclr_project_pack (projection + L2-norm + pack in one launch) instead of
    crossclr_normalize, and the step behaves like prenormalized=True from there on (projection.py)."""
    import torch.distributed as dist
    lib = nat.library()
    dev = video.device
    b, D = video.shape
    if project is not None:
        D = int(project[5])
    world = dist.get_world_size(group) if group is not None else 1
    rank = dist.get_rank(group) if group is not None else 0
    # CROSSCLR_FORCE_SHARDED_PATH=1 (test knob): take the multi-rank code path (all-gather, second launch
    # with skip_rank, statistics gather, loss all-reduce) even for a 1-rank group, so the collectives and the
    # skip logic can be exercised on a single GPU
    sharded = world > 1 or (group is not None and os.environ.get("CROSSCLR_FORCE_SHARDED_PATH") == "1")
    small_tau = bool(lib.crossclr_needs_row_shift(float(temperature), float(negative_w)))
    mode = _resolve_mode(compute_mode, b * world, video.dtype, small_tau)
    plan = _plan_for(b, D, world, rank, mode)
    if not sharded and project is None:
        return _forward_step(video, text, temperature, negative_w, plan, mode, negative_scale, loss_weight, save_for_backward, prenormalized)
    stash_everywhere = True
    if world > 1:
        # (what a rank of the pair scheme keeps alive between forward and backward: the local stash and one rectangular stash per block
        #  it evaluates itself -- K pair partners and the antipode)
        need = 0
        if save_for_backward and plan.stash_bytes > 0 and plan.fast_path == 1:
            nblocks = (world - 1) // 2 + (1 if world % 2 == 0 else 0)
            need = int(plan.stash_bytes + nblocks * lib.crossclr_rect_stash_bytes(ctypes.byref(plan), 1))
        stash_everywhere = _check_equal_rows_per_rank(b, D, group, dev, need)
    stream = _stream_for(video)
    f32 = dict(dtype=torch.float32, device=dev)

    ws = _Workspace()
    ws.step = None
    ws.plan, ws.world, ws.rank, ws.sharded = plan, world, rank, sharded
    ws.temperature, ws.negative_w = float(temperature), float(negative_w)
    ws.in_dtype = _IN_DTYPE[video.dtype]
    # one allocation for everything the step keeps (a dozen torch.empty calls were a third of the host time at small batches):
    # [xhat | loss_sum (doubles) | inv_norm | diag | logz | rz | wrz | part], every piece 256-byte aligned
    n2 = 2 * plan.bpad
    sizes = [plan.operand_bytes, 0, 4 * n2, 4 * plan.bpad, 4 * n2, 4 * n2, 4 * n2, 4 * plan.fwd_ws_floats]
    offs, tot = [], 0
    for sz in sizes:
        offs.append(tot)
        tot += (sz + 255) // 256 * 256
    slab = torch.empty(tot, dtype=torch.uint8, device=dev)
    ws.xhat = slab[offs[0]:offs[0] + sizes[0]]
    # (the loss the caller gets back is a 0-dim view of this buffer: its own small allocation, so it never pins the slab)
    ws.loss_sum = torch.empty(max(2, plan.loss_ws_doubles), dtype=torch.float64, device=dev)
    ws.inv_norm = _carve(slab, offs[2], sizes[2], torch.float32)
    ws.diag = _carve(slab, offs[3], sizes[3], torch.float32)
    ws.logz = _carve(slab, offs[4], sizes[4], torch.float32)
    ws.rz = _carve(slab, offs[5], sizes[5], torch.float32)
    ws.wrz = _carve(slab, offs[6], sizes[6], torch.float32)
    part = _carve(slab, offs[7], sizes[7], torch.float32)
    nlaunch = 2 if sharded else 1
    pp = ctypes.byref(plan)
    ws.group, ws.partner_peers = group, None
    partner_possible = False
    ws.k_rows = _pack_pair(negative_scale, b, plan.bpad, dev, "negative_scale")
    ws.lw = _pack_pair(loss_weight, b, plan.bpad, dev, "loss_weight")
    ws.k_cols = ws.k_rows

    ws.prenormalized = bool(prenormalized) or project is not None
    ws.xf = ws.xf_all = None
    if project is not None:
        if plan.fast_path != 1 or plan.Dpad > 1024 or mode != nat.MODE_BF16:
            raise RuntimeError("the fused projection needs the bf16 register-resident path (embed_dim <= 1024)")
        wv, wt, ldws, bias_v, bias_t, _ = project
        with _Range("crossclr.project_pack"):
            nat.check(lib.crossclr_project_pack_wf(pp, _ptr(video), _ptr(text), video.stride(0), text.stride(0), video.shape[1], text.shape[1],
                                                ws.in_dtype, _ptr(wv), _ptr(wt), ldws[0], ldws[1], _ptr(bias_v), _ptr(bias_t), _ptr(ws.xhat),
                                                _ptr(ws.inv_norm), _ptr(ws.diag), stream))
            # the saved backward's fragment-major copy, re-laid from the packed rows (16 MiB at b = 8192, D = 512: a few microseconds
            # against the ~10 % the fragment-major backward is faster by)
            if save_for_backward and plan.xf_bytes > 0 and plan.stash_bytes > 0 and not small_tau and _use_xf(plan) and world == 1:
                ws.xf = torch.empty(plan.xf_bytes, dtype=torch.uint8, device=dev)
                nat.check(lib.crossclr_pack_xf_from_packed(pp, ws.xhat.data_ptr(), 1, ws.xf.data_ptr(), stream))
    else:
        # With a saved backward to follow (plan.xf_bytes > 0: bf16 register-resident path, D <= 512) the same launch also leaves the
        # unit rows in the fragment-major layout that backward loads straight into MFMA fragments (crossclr_backward_saved_xf).
        use_xf = save_for_backward and plan.xf_bytes > 0 and plan.stash_bytes > 0 and not small_tau and _use_xf(plan)
        ws.xf = torch.empty(plan.xf_bytes, dtype=torch.uint8, device=dev) if use_xf else None
        with _Range("crossclr.normalize"):
            if ws.xf is not None:
                entry = lib.crossclr_pack_xf if prenormalized else lib.crossclr_normalize_xf
                nat.check(entry(pp, _ptr(video), _ptr(text), video.stride(0), text.stride(0), ws.in_dtype,
                                _ptr(ws.xhat), _ptr(ws.xf), _ptr(ws.inv_norm), _ptr(ws.diag), stream))
            else:
                entry = lib.crossclr_pack if prenormalized else lib.crossclr_normalize    # unit rows are only laid out, not re-normalised
                nat.check(entry(pp, _ptr(video), _ptr(text), video.stride(0), text.stride(0), ws.in_dtype,
                                _ptr(ws.xhat), _ptr(ws.inv_norm), _ptr(ws.diag), stream))
    gather = None
    if sharded:
        # all-gather of the packed operands runs on the collective's own stream (RCCL over xGMI)
        # while the local column block is processed on the compute stream
        ws.xcols = torch.empty(world * plan.operand_bytes, dtype=torch.uint8, device=dev)
        first_peers = None
        # Partner gradients (default with the pair scheme and a backward to follow): the rank that evaluated a pair block also forms
        # that block's contribution to its partner's gradient from the saved exponentials and ships it, instead of the partner
        # recomputing the block.  Decided from rank-invariant facts only (every rank must agree on what travels).
        partner_possible = (save_for_backward and world >= 3 and plan.fast_path == 1 and plan.stash_bytes > 0 and not small_tau and stash_everywhere and
                            os.environ.get("CROSSCLR_DISABLE_PAIR_FORWARD") != "1" and os.environ.get("CROSSCLR_DISABLE_REMOTE_SAVE") != "1" and
                            os.environ.get("CROSSCLR_PARTNER_GRADS", "1") != "0" and lib.crossclr_rect_stash_bytes(pp, 1) > 0)
        pairs_apply = world >= 3 and plan.fast_path == 1 and not small_tau and os.environ.get("CROSSCLR_DISABLE_PAIR_FORWARD") != "1"
        xmode = _exchange_mode(world, pairs_apply)
        if xmode in ("p2p", "p2p_each") and pairs_apply:
            first_peers = [(rank + 1 + k) % world for k in range((world - 1) // 2)]     # in the order the forward needs them
            if world % 2 == 0:
                first_peers.append((rank + world // 2) % world)
        gather = ws.exchange = _OperandExchange(ws.xcols, ws.xhat, group, world, rank, first_peers, each=(xmode == "p2p_each"),
                                                defer_late=partner_possible)
        global _last_exchange_mode
        _last_exchange_mode = gather.mode
        ws.k_work = None
        if ws.k_rows is not None:   # (asynchronous: the local block does not need the other ranks' negative scales)
            ws.k_cols = torch.empty(world * ws.k_rows.numel(), **f32)
            ws.k_work = dist.all_gather_into_tensor(ws.k_cols, ws.k_rows, group=group, async_op=True)
    else:
        ws.xcols = ws.xhat
        ws.exchange = None
        ws.k_work = None

    def k_cols_ready():
        if ws.k_work is not None:
            _traced_wait("negative scales", [ws.k_work])
            ws.k_work = None
    # Small temperatures (max |logit| = max(1,|w|)/tau > 128): no single soft-max shift fits fp32.  Like the reference's
    # float64 soft-max (loss.py:60) the rows then get their own shift -- the row maximum, found by a first pass -- and the
    # generic tiled kernels do the rest (no symmetric evaluation, no pair scheme, no save-for-backward in this regime).
    ws.shift = ws.shift_cols = None
    if lib.crossclr_needs_row_shift(ws.temperature, ws.negative_w):
        return _forward_row_shift(lib, ws, part, gather, group, stream, b, world, rank, dev, save_for_backward, k_cols_ready)
    # Local (symmetric) block.  When a backward will follow and the plan offers it, the forward also saves its bf16
    # exponentials (plan.stash_bytes, 0.27 GB at b = 8192) so that the backward does not recompute the similarity
    # product -- the analogue of the reference's autograd-saved [B,2B] float64 tensors, 50x smaller.
    ws.stash = _alloc_stash(plan.stash_bytes, dev) if save_for_backward else None
    if ws.stash is None:
        ws.xf = None      # (no saved backward after all: the recomputing one reads the row-major operand)
    if partner_possible and ws.stash is None:
        raise RuntimeError("CrossCLR (sharded): could not allocate the saved-exponentials buffer on this rank; the ranks agreed on the "
                           "partner-gradient scheme, which needs it -- set CROSSCLR_PARTNER_GRADS=0 (or CROSSCLR_DISABLE_SAVE=1) on every rank")
    with _Range("crossclr.forward"):
        if ws.stash is not None:
            nat.check(lib.crossclr_forward_save(pp, _ptr(ws.xhat), ws.temperature, ws.negative_w,
                                                _sw(ws.k_rows, ws.k_rows, None), _ptr(part), 0, _ptr(ws.stash), stream))
        else:
            nat.check(lib.crossclr_forward_w(pp, _ptr(ws.xhat), _ptr(ws.xhat), 1, rank, -1, ws.temperature, ws.negative_w,
                                             _sw(ws.k_rows, ws.k_rows, None), _ptr(part), 0, stream))
    # Pair evaluation (bf16 register-resident path, >= 3 ranks): the (r, s) block of the symmetric matrix of exponentials
    # is evaluated by ONE of the two ranks, which ships its column sums to the other -- 3 instead of 7 remote blocks at
    # 8 ranks (+ the antipodal one, evaluated by both).  CROSSCLR_DISABLE_PAIR_FORWARD=1: every rank evaluates all blocks.
    use_pairs = (sharded and plan.fast_path == 1 and world >= 3 and
                 os.environ.get("CROSSCLR_DISABLE_PAIR_FORWARD") != "1")
    # With a backward to follow, the remote blocks this rank evaluates itself (pair partners, antipodal rank; at 2 ranks: the
    # other rank) save their exponentials as well: the backward then recomputes only the blocks the OTHER ranks evaluated.
    ws.saved_blocks, ws.recompute_ranges = None, None
    save_remote = (sharded and world >= 2 and ws.stash is not None and plan.fast_path == 1 and (use_pairs or world == 2) and
                   lib.crossclr_rect_stash_bytes(pp, 1) > 0 and   # (0 beyond D = 1024: the register-resident kernels end there)
                   os.environ.get("CROSSCLR_DISABLE_REMOTE_SAVE") != "1")
    npairs = (world - 1) // 2 if use_pairs else 0
    if sharded and (save_remote or use_pairs):
        n2 = 2 * plan.bpad
        k_cols_ready()
        sw_all = _sw(ws.k_rows, ws.k_cols, None)
        peers = [(rank + 1 + k) % world for k in range(npairs)]
        opp = (rank + world // 2) % world if world % 2 == 0 else None    # the antipodal rank (at 2 ranks: the other rank)
        if save_remote:
            ws.saved_blocks, ws.recompute_ranges = [], []
        if npairs:
            outbox, inbox, in_split, out_split, out_row = _pair_exchange_buffers(dev, group, stream, world, rank, n2, npairs)
        # One launch per peer when the operands arrive peer by peer (p2p_each) and the workspace has a launch group for each
        # (local + peers + antipode + received <= 8); otherwise ONE launch over the whole pair range behind one wait.
        per_peer = npairs > 0 and gather.mode == "p2p_each" and npairs + 3 <= nat.LAUNCH_GROUPS
        group_of = 1
        if npairs and per_peer:
            for k, peer in enumerate(peers):
                gather.wait_peer(peer)
                cs = outbox[out_row[peer]]
                if save_remote:
                    st = _alloc_rect_stash(lib.crossclr_rect_stash_bytes(pp, 1), dev)
                    nat.check(lib.crossclr_forward_rect_save(pp, _ptr(ws.xhat), _ptr(ws.xcols), peer, 1, 1, ws.temperature, ws.negative_w,
                                                             sw_all, _ptr(part), group_of * plan.fwd_slots, _ptr(cs), _ptr(st), stream))
                    ws.saved_blocks.append((peer, 1, st))
                else:
                    nat.check(lib.crossclr_forward_pairs(pp, _ptr(ws.xhat), _ptr(ws.xcols), peer, 1, ws.temperature, ws.negative_w,
                                                         sw_all, _ptr(part), group_of * plan.fwd_slots, _ptr(cs), stream))
                group_of += 1
        elif npairs:
            gather.wait_forward()   # (p2p: the peers this rank evaluates itself; all-gather: everything)
            for peer in peers:      # (p2p_each beyond the workspace's launch groups: the slices arrive one by one, the launch needs all)
                gather.wait_peer(peer)
            colsum = torch.empty(npairs * n2, **f32)
            if save_remote:
                st = _alloc_rect_stash(lib.crossclr_rect_stash_bytes(pp, npairs), dev)
                nat.check(lib.crossclr_forward_rect_save(pp, _ptr(ws.xhat), _ptr(ws.xcols), peers[0], npairs, 1, ws.temperature,
                                                         ws.negative_w, sw_all, _ptr(part), plan.fwd_slots, _ptr(colsum), _ptr(st), stream))
                ws.saved_blocks.append((peers[0], npairs, st))
            else:
                nat.check(lib.crossclr_forward_pairs(pp, _ptr(ws.xhat), _ptr(ws.xcols), peers[0], npairs, ws.temperature,
                                                     ws.negative_w, sw_all, _ptr(part), plan.fwd_slots, _ptr(colsum), stream))
            # the outbox rows are ordered by peer rank (the all-to-all's order), the launch's by distance: one small copy per peer
            cv = colsum.view(npairs, n2)
            for k, peer in enumerate(peers):
                outbox[out_row[peer]].copy_(cv[k])
            group_of = 2
        if save_remote and npairs:
            if partner_possible:
                ws.partner_peers = peers        # the backward ships these blocks' transposed contributions instead
            else:
                ws.recompute_ranges.append(((rank - npairs) % world, npairs))
        if opp is not None:   # both sides evaluate their own rows of the antipodal block
            gather.wait_peer(opp)
            if save_remote:
                st = _alloc_rect_stash(lib.crossclr_rect_stash_bytes(pp, 1), dev)
                nat.check(lib.crossclr_forward_rect_save(pp, _ptr(ws.xhat), _ptr(ws.xcols), opp, 1, 0, ws.temperature, ws.negative_w,
                                                         sw_all, _ptr(part), group_of * plan.fwd_slots, None, _ptr(st), stream))
                ws.saved_blocks.append((opp, 1, st))
            else:
                sw_opp = None
                if ws.k_rows is not None:
                    sw_opp = ctypes.pointer(nat.SampleWeights(ws.k_rows.data_ptr(), ws.k_cols.data_ptr() + 4 * opp * n2, 0))
                nat.check(lib.crossclr_forward_w(pp, _ptr(ws.xhat), ctypes.c_void_p(ws.xcols.data_ptr() + opp * plan.operand_bytes),
                                                 1, opp, -1, ws.temperature, ws.negative_w, sw_opp, _ptr(part),
                                                 group_of * plan.fwd_slots, stream))
            group_of += 1
        if npairs:
            # ship the column sums of block (r, s) to rank s: ONE all-to-all that moves exactly the rows somebody is owed
            # (npairs x 2 bpad floats out, as many in; the received rows are added in rank order: deterministic)
            dist.all_to_all_single(inbox.view(-1), outbox.view(-1), output_split_sizes=out_split, input_split_sizes=in_split, group=group)
            received = inbox.sum(0) if npairs > 1 else inbox[0]     # (kept in a variable: _ptr is a bare address)
            nat.check(lib.crossclr_forward_add(pp, _ptr(part), group_of * plan.fwd_slots, _ptr(received), stream))
            group_of += 1
        nlaunch = group_of
    elif sharded:
        gather.wait()
        k_cols_ready()
        # wide bf16 plans (D > 1024) with a backward to follow: the block against the other world - 1 ranks saves its bf16 records too (the
        # generic forward in the rectangular layout the D-slice backward reads: 0.5 GiB per rank-block at b = 8192);
        # exact-fp32 plans with a backward to follow: the block against the other world - 1 ranks saves its fp32 exponentials too
        # (crossclr_forward_rect_save / crossclr_backward_rect_saved on the generic kernels: 1 GiB per rank-block at b = 8192), so that
        # the backward of the remote blocks is the gradient product alone (2.3 instead of 6.8 ms per 8192^2 block)
        rect_bytes = 0
        if ((mode == nat.MODE_FP32 or plan.Dpad > 1024) and world >= 2 and ws.stash is not None and stash_everywhere and
                os.environ.get("CROSSCLR_DISABLE_REMOTE_SAVE") != "1"):
            rect_bytes = int(lib.crossclr_rect_stash_bytes(pp, world - 1))
        st_r = _alloc_stash(rect_bytes, dev) if rect_bytes > 0 else None
        if st_r is not None:
            first = (rank + 1) % world
            nat.check(lib.crossclr_forward_rect_save(pp, _ptr(ws.xhat), _ptr(ws.xcols), first, world - 1, 0, ws.temperature, ws.negative_w,
                                                     _sw(ws.k_rows, ws.k_cols, None), _ptr(part), plan.fwd_slots, None, _ptr(st_r), stream))
            ws.saved_blocks, ws.recompute_ranges = [(first, world - 1, st_r)], []
        else:
            nat.check(lib.crossclr_forward_w(pp, _ptr(ws.xhat), _ptr(ws.xcols), world, 0, rank, ws.temperature,
                                             ws.negative_w, _sw(ws.k_rows, ws.k_cols, None), _ptr(part), plan.fwd_slots, stream))
    with _Range("crossclr.forward_finish"):
        nat.check(lib.crossclr_forward_finish_w(pp, _ptr(part), nlaunch * plan.fwd_slots, _ptr(ws.diag), ws.temperature,
                                                ws.negative_w, _sw(ws.k_rows, ws.k_rows, ws.lw), _ptr(ws.logz), _ptr(ws.rz),
                                                _ptr(ws.wrz), _ptr(ws.loss_sum), stream))
    if sharded:
        # the backward's remote launch needs every rank's omega/Z: gathered asynchronously, waited for in the backward
        # (w * omega/Z of the columns is recomputed there: the same fp32 product the finish kernel forms).  Only when a
        # backward will follow: under no_grad nobody would ever wait for the collective or keep its buffer alive.
        ws.rz_cols, ws.wrz_cols, ws.stats_work = None, None, None
        if save_for_backward:
            ws.rz_cols = torch.empty(world * ws.rz.numel(), **f32)
            ws.stats_work = dist.all_gather_into_tensor(ws.rz_cols, ws.rz, group=group, async_op=True)
        total = ws.loss_sum[:1].clone()
        dist.all_reduce(total, group=group)
        loss = (total / (2.0 * b * world)).reshape(())
        if not save_for_backward and ws.exchange is not None:
            ws.exchange.wait()     # nobody will wait later: the late slices' sends / receives must not outlive their buffers
        elif partner_possible and ws.partner_peers is not None:
            ws.exchange.finish()   # (deferred late slices are never requested in this scheme)
    else:
        ws.rz_cols, ws.wrz_cols, ws.stats_work = ws.rz, ws.wrz, None
        loss = ws.loss_sum[1]      # = sum / (2 B), written by the finish kernel (a 0-dim view of the 8-byte-per-block loss buffer)
    return loss, ws


def _forward_row_shift(lib, ws, part, gather, group, stream, b, world, rank, dev, needs_backward, k_cols_ready):
    """Two-pass forward (include/crossclr.h, "two-pass soft-max"): row maxima, then sums relative to them.
    k_cols_ready: waits for the asynchronous gather of the other ranks' negative scales (ws.k_cols) -- every launch over the
    gathered operand reads them, here and in the backward (`gather.wait()` covers the operand exchange only)."""
    import torch.distributed as dist
    plan = ws.plan
    pp = ctypes.byref(plan)
    f32 = dict(dtype=torch.float32, device=dev)
    ws.stash = None
    ws.shift = torch.empty(2 * plan.bpad, **f32)
    sw_loc, sw_all = _sw(ws.k_rows, ws.k_rows, None), _sw(ws.k_rows, ws.k_cols, None)
    T, w = ws.temperature, ws.negative_w
    nat.check(lib.crossclr_forward_rowmax(pp, _ptr(ws.xhat), _ptr(ws.xhat), 1, rank, -1, T, w, sw_loc, _ptr(part), _ptr(ws.shift), 0,
                                          stream))
    if ws.sharded:
        gather.wait()
        k_cols_ready()
        nat.check(lib.crossclr_forward_rowmax(pp, _ptr(ws.xhat), _ptr(ws.xcols), world, 0, rank, T, w, sw_all, _ptr(part),
                                              _ptr(ws.shift), 1, stream))
    # sharded run with a backward to follow (exact-fp32 plans: fp32 fragments; bf16 register-resident plans: bf16 records): the block against
    # the other world - 1 ranks saves U AND Ut (relative to the remote rows' shifts), so the ranks exchange their row maxima HERE, between the passes (2 bpad floats per rank; every rank takes this
    # branch or none: the condition reads the plan and the environment only).  The backward then recomputes nothing at any temperature.
    ws.saved_blocks, ws.recompute_ranges = None, None
    shift_all = None
    if (ws.sharded and needs_backward and os.environ.get("CROSSCLR_DISABLE_REMOTE_SAVE") != "1" and
            int(lib.crossclr_rect_stash_bytes_s(pp, world - 1)) > 0 and int(lib.crossclr_stash_bytes_s(pp)) > 0):
        shift_all = torch.empty(world * ws.shift.numel(), **f32)
        dist.all_gather_into_tensor(shift_all, ws.shift, group=group)
    # second pass over the local block; with a backward to follow (exact-fp32 plans) it also saves the exponentials relative to the
    # row's and to the column's shift, and the backward does not recompute the similarity product
    ws.stash = _alloc_stash(lib.crossclr_stash_bytes_s(pp), dev) if needs_backward else None
    if ws.stash is not None:
        nat.check(lib.crossclr_forward_save_s(pp, _ptr(ws.xhat), T, w, sw_loc, _ptr(ws.shift), _ptr(part), 0, _ptr(ws.stash), stream))
    else:
        nat.check(lib.crossclr_forward_s(pp, _ptr(ws.xhat), _ptr(ws.xhat), 1, rank, -1, T, w, sw_loc, _ptr(ws.shift), _ptr(part), 0, stream))
    nlaunch = 1
    if ws.sharded:
        st_r = None
        if shift_all is not None and ws.stash is not None:
            st_r = _alloc_stash(int(lib.crossclr_rect_stash_bytes_s(pp, world - 1)), dev)
        if st_r is not None:
            first = (rank + 1) % world
            nat.check(lib.crossclr_forward_rect_save_s(pp, _ptr(ws.xhat), _ptr(ws.xcols), first, world - 1, T, w, sw_all, _ptr(ws.shift),
                                                       _ptr(shift_all), _ptr(part), plan.fwd_slots, _ptr(st_r), stream))
            ws.saved_blocks = [(first, world - 1, st_r)]
        else:
            nat.check(lib.crossclr_forward_s(pp, _ptr(ws.xhat), _ptr(ws.xcols), world, 0, rank, T, w, sw_all, _ptr(ws.shift), _ptr(part),
                                             plan.fwd_slots, stream))
        nlaunch = 2
    nat.check(lib.crossclr_forward_finish_s(pp, _ptr(part), nlaunch * plan.fwd_slots, _ptr(ws.diag), T, w,
                                            _sw(ws.k_rows, ws.k_rows, ws.lw), _ptr(ws.shift), _ptr(ws.logz), _ptr(ws.rz), _ptr(ws.wrz),
                                            _ptr(ws.loss_sum), stream))
    if ws.sharded:
        # the remote backward needs every rank's omega/Z' AND the shifts they are relative to: one gather of both
        ws.rz_cols, ws.wrz_cols, ws.shift_cols, ws.stats_work = None, None, None, None
        if needs_backward:
            both = torch.cat([ws.rz, ws.shift])
            gathered = torch.empty(world * both.numel(), **f32)
            ws.stats_work = dist.all_gather_into_tensor(gathered, both, group=group, async_op=True)
            ws.rz_cols = gathered   # split after the wait, in the backward
        total = ws.loss_sum[:1].clone()
        dist.all_reduce(total, group=group)
        loss = (total / (2.0 * b * world)).reshape(())
    else:
        ws.rz_cols, ws.wrz_cols, ws.shift_cols, ws.stats_work = ws.rz, ws.wrz, ws.shift, None
        loss = ws.loss_sum[1]
    return loss, ws


def _xf_selftest_remote(dev, D: int, weighted: bool, launches: int = 2):
    """The pair kernel's RECTANGULAR and TRANSPOSED instantiations (crossclr_backward_rect_saved_xfp / _t_xfp: separate template modes with their
    own hand-counted waits) against the LDS-staged crossclr_backward_rect_saved / _t over the same saved exponentials, bit for bit: rank 1 of a
    synthetic 3-rank run (384 rows per rank: one pair block, wrap-around of the rank segments).  True / False, None when there is nothing to
    compare on this build."""
    lib = nat.library()
    stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
    world, rank, b = 3, 1, 384
    plan = nat.make_plan(b, D, world, rank, nat.MODE_BF16)
    pp = ctypes.byref(plan)
    nb = lib.crossclr_rect_stash_bytes(pp, 1)
    if not (plan.xf_bytes > 0 and nb > 0):
        return None
    f32 = dict(dtype=torch.float32, device=dev)
    g = torch.Generator().manual_seed(77)
    n2 = 2 * plan.bpad
    xall = torch.empty(world * plan.operand_bytes, dtype=torch.uint8, device=dev)
    xf_all = torch.empty(world * plan.operand_bytes, dtype=torch.uint8, device=dev)
    inv, diag = torch.empty(n2, **f32), torch.empty(plan.bpad, **f32)
    for r in range(world):
        v = torch.randn(b, D, generator=g)
        t = (v + 0.5 * torch.randn(b, D, generator=g)).to(dev)
        v = v.to(dev)
        nat.check(lib.crossclr_normalize(pp, _ptr(v), _ptr(t), v.stride(0), t.stride(0), nat.IN_F32, xall.data_ptr() + r * plan.operand_bytes,
                                         _ptr(inv), _ptr(diag), stream))
    nat.check(lib.crossclr_pack_xf_from_packed(pp, _ptr(xall), world, _ptr(xf_all), stream))
    xr = xall[rank * plan.operand_bytes:(rank + 1) * plan.operand_bytes]
    xfr = xf_all[rank * plan.operand_bytes:(rank + 1) * plan.operand_bytes]
    k_all = None
    if weighted:
        k_all = (torch.rand(world * n2, generator=g) > 0.3).float().to(dev)
    k_rows = None if k_all is None else k_all[rank * n2:(rank + 1) * n2]
    sw = _sw(k_rows, k_all, None)
    part = torch.empty(plan.fwd_ws_floats, **f32)
    colsum = torch.empty(n2, **f32)
    st = torch.empty(nb, dtype=torch.uint8, device=dev)
    first = (rank + 1) % world
    nat.check(lib.crossclr_forward_rect_save(pp, _ptr(xr), _ptr(xall), first, 1, 1, 0.05, 0.8, sw, _ptr(part), plan.fwd_slots, _ptr(colsum), _ptr(st),
                                             stream))
    # any positive statistics will do: the comparison is kernel against kernel on the same inputs
    rz_all = (torch.rand(world * n2, generator=g) * 1e-3 + 1e-4).to(dev)
    wrz_all = 0.8 * rz_all
    rz, wrz = rz_all[rank * n2:(rank + 1) * n2], wrz_all[rank * n2:(rank + 1) * n2]
    want = torch.empty(plan.gbuf_bytes // 4, **f32)
    want_t = torch.empty(plan.gbuf_bytes // 4, **f32)
    nat.check(lib.crossclr_backward_rect_saved(pp, _ptr(xall), _ptr(st), first, 1, 0.05, 0.8, _ptr(rz), _ptr(wrz), _ptr(rz_all), _ptr(wrz_all), sw,
                                               _ptr(want), 0, stream))
    nat.check(lib.crossclr_backward_rect_saved_t(pp, _ptr(xr), _ptr(st), first, 1, 0, 0.05, 0.8, _ptr(rz), _ptr(wrz), _ptr(rz_all), _ptr(wrz_all), sw,
                                                 _ptr(want_t), stream))
    ok = True
    for _ in range(launches):
        got, got_t = torch.full_like(want, float("nan")), torch.full_like(want_t, float("nan"))
        nat.check(lib.crossclr_backward_rect_saved_xfp(pp, _ptr(xf_all), _ptr(st), first, 1, 0.05, 0.8, _ptr(rz), _ptr(wrz), _ptr(rz_all), _ptr(wrz_all),
                                                       sw, _ptr(got), 0, stream))
        nat.check(lib.crossclr_backward_rect_saved_t_xfp(pp, _ptr(xfr), _ptr(st), first, 1, 0, 0.05, 0.8, _ptr(rz), _ptr(wrz), _ptr(rz_all),
                                                         _ptr(wrz_all), sw, _ptr(got_t), stream))
        ok = ok and bool(torch.equal(got, want)) and bool(torch.equal(got_t, want_t))
    return ok


def _remote_xfp(ws, plan, lib, pp, dev) -> bool:
    """Do the saved backwards of the REMOTE blocks (and the partner gradients) run the pair kernel on fragment-major operands?  Needs the
    local fragment-major copy (ws.xf: the step took the fragment-major pair), the pair kernel verified on this device for the local
    block, and a gathered operand / rectangular stashes below 4 GiB (32-bit offsets).  CROSSCLR_REMOTE_XFP=0 keeps the LDS-staged kernels."""
    if ws.xf is None or os.environ.get("CROSSCLR_REMOTE_XFP", "1") == "0":
        return False
    if plan.Dpad > 1024:      # wide plans: the rectangular launches exist on the LDS-staged kernel only
        return False
    if ws.world * plan.operand_bytes >= (1 << 32):
        return False
    if any(lib.crossclr_rect_stash_bytes(pp, n) >= (1 << 32) for _, n, _ in (ws.saved_blocks or [])):
        return False
    if _saved_backward_entry(ws, plan, dev) != "crossclr_backward_saved_xfp":
        return False
    # the rectangular and transposed instantiations are verified on their own (round-4 review: the local block's self-test says nothing about them)
    weighted = ws.k_rows is not None
    key = (str(dev), plan.Dpad, weighted, "crossclr_backward_rect_saved_xfp", nat.library_path())
    ok = _xf_verified.get(key)
    if ok is None:
        if nat.injected_for_testing():
            ok = True
        elif dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return False
        else:
            verdict = _xf_selftest_remote(dev, plan.D, weighted)
            if verdict is False:
                warnings.warn(f"CrossCLR: crossclr_backward_rect_saved_xfp / _t_xfp disagree with the LDS-staged kernels on this device / build at "
                              f"Dpad = {plan.Dpad}; remote blocks keep the LDS-staged kernels in this process")
            ok = bool(verdict)
        _xf_verified[key] = ok
    return ok


def _backward_impl(ws: _Workspace, video: torch.Tensor, text: torch.Tensor, grad_out: torch.Tensor):
    if ws.step is not None and ws.prenormalized != 2:
        return _backward_step(ws, video, text, grad_out)
    lib = nat.library()
    plan = ws.plan
    pp = ctypes.byref(plan)
    dev = video.device
    stream = _stream_for(video)
    gbuf = torch.empty(plan.gbuf_bytes // 4, dtype=torch.float32, device=dev)
    rank, world = ws.rank, ws.world
    rz_loc, wrz_loc = ws.rz, ws.wrz   # column statistics of the local block = this rank's row statistics
    if ws.shift is not None:   # two-pass (small temperature) regime: generic kernels with per-row shifts
        if ws.stash is not None:
            nat.check(lib.crossclr_backward_saved_s(pp, _ptr(ws.xhat), _ptr(ws.stash), ws.temperature, ws.negative_w, _ptr(ws.rz),
                                                    _ptr(ws.wrz), _sw(ws.k_rows, ws.k_rows, None), _ptr(gbuf), 0, stream))
            ws.stash = None
        else:
            nat.check(lib.crossclr_backward_s(pp, _ptr(ws.xhat), _ptr(ws.xhat), 1, rank, -1, ws.temperature, ws.negative_w,
                                              _ptr(ws.rz), _ptr(ws.wrz), _ptr(ws.rz), _ptr(ws.wrz), _sw(ws.k_rows, ws.k_rows, None),
                                              _ptr(ws.shift), _ptr(ws.shift), _ptr(gbuf), 0, stream))
        if ws.sharded:
            if ws.wrz_cols is None:
                ws.stats_work.wait()
                n2 = ws.rz.numel()
                both = ws.rz_cols.view(world, 2, n2)
                ws.rz_cols = both[:, 0].contiguous().view(-1)
                ws.shift_cols = both[:, 1].contiguous().view(-1)
                ws.wrz_cols = ws.rz_cols * ws.negative_w
            if ws.saved_blocks:   # exact-fp32 plans: U and Ut of the block against the other ranks were saved (crossclr_forward_rect_save_s)
                for first, n, st in ws.saved_blocks:
                    nat.check(lib.crossclr_backward_rect_saved_s(pp, _ptr(ws.xcols), _ptr(st), first, n, ws.temperature, ws.negative_w,
                                                                 _ptr(ws.rz), _ptr(ws.wrz), _ptr(ws.rz_cols), _ptr(ws.wrz_cols),
                                                                 _sw(ws.k_rows, ws.k_cols, None), _ptr(gbuf), 1, stream))
                ws.saved_blocks = None
            else:
                nat.check(lib.crossclr_backward_s(pp, _ptr(ws.xhat), _ptr(ws.xcols), world, 0, rank, ws.temperature, ws.negative_w,
                                                  _ptr(ws.rz), _ptr(ws.wrz), _ptr(ws.rz_cols), _ptr(ws.wrz_cols),
                                                  _sw(ws.k_rows, ws.k_cols, None), _ptr(ws.shift), _ptr(ws.shift_cols), _ptr(gbuf), 1, stream))
    elif ws.stash is not None:
        with _Range("crossclr.backward"):
            entry = _saved_backward_entry(ws, plan, dev)
            operand = ws.xhat if entry == "crossclr_backward_saved" else ws.xf
            nat.check(getattr(lib, entry)(pp, _ptr(operand), _ptr(ws.stash), ws.temperature, ws.negative_w, _ptr(ws.rz), _ptr(ws.wrz),
                                          _sw(ws.k_rows, ws.k_rows, None), _ptr(gbuf), 0, stream))
        ws.stash = None   # consumed: give the 0.27 GB back to the allocator as soon as the launch is queued
        if not (ws.sharded and ws.saved_blocks is not None):
            ws.xf = None      # (a sharded step's partner gradients read it once more below)
    else:
        nat.check(lib.crossclr_backward_w(pp, _ptr(ws.xhat), _ptr(ws.xhat), 1, rank, -1, ws.temperature, ws.negative_w,
                                          _ptr(ws.rz), _ptr(ws.wrz), _ptr(rz_loc), _ptr(wrz_loc),
                                          _sw(ws.k_rows, ws.k_rows, None), _ptr(gbuf), 0, stream))
    if ws.sharded and ws.shift is None and ws.saved_blocks is not None:
        if ws.wrz_cols is None:
            _traced_wait("statistics", [ws.stats_work])
            ws.wrz_cols = ws.rz_cols * ws.negative_w
        sw_all = _sw(ws.k_rows, ws.k_cols, None)
        partner_work = None
        # remote blocks on the fragment-major operand: the slices this rank received are re-laid out HERE (the exchange moved the row-major
        # operand once; crossclr_pack_xf_from_packed: the slices of the blocks this rank evaluated, ~40 us for 117 MB at 8 ranks), the
        # transposes read this rank's own copy
        remote_xfp = _remote_xfp(ws, plan, lib, pp, dev)
        if remote_xfp:
            ws.xf_all = torch.empty(world * plan.operand_bytes, dtype=torch.uint8, device=dev)
            for first, n, _ in ws.saved_blocks:
                runs = [(first, min(n, world - first))] + ([(0, n - (world - first))] if first + n > world else [])
                for r0, cnt in runs:
                    nat.check(lib.crossclr_pack_xf_from_packed(pp, ws.xcols.data_ptr() + r0 * plan.operand_bytes, cnt,
                                                               ws.xf_all.data_ptr() + r0 * plan.operand_bytes, stream))
        if ws.partner_peers:
            # Pair blocks, the partner's half first (its transfer then hides behind this rank's own blocks): block (r, s) transposed,
            # from the exponentials saved for it -> [2 bpad, Dpad] fp32 per partner, shipped with one all-to-all.
            import torch.distributed as dist
            n2, npairs = 2 * plan.bpad, len(ws.partner_peers)
            nel = n2 * plan.Dpad
            _, _, in_split, out_split, out_row = _pair_exchange_buffers(dev, ws.group, stream, world, rank, n2, npairs)
            outg, ing, tmp = _partner_gradient_buffers(dev, ws.group, stream, npairs, nel, plan.gbuf_bytes // 4)
            for first, n, st in ws.saved_blocks:
                for which in range(n):
                    peer = (first + which) % world
                    if peer not in out_row or peer not in ws.partner_peers:
                        continue                                # (the antipodal block: both sides own their rows)
                    entry_t, operand_t = ((lib.crossclr_backward_rect_saved_t_xfp, ws.xf) if remote_xfp else
                                          (lib.crossclr_backward_rect_saved_t, ws.xhat))
                    nat.check(entry_t(pp, _ptr(operand_t), _ptr(st), first, n, which, ws.temperature, ws.negative_w,
                                      _ptr(ws.rz), _ptr(ws.wrz), _ptr(ws.rz_cols), _ptr(ws.wrz_cols), sw_all, _ptr(tmp), stream))
                    torch.sum(tmp.view(-1, nel), 0, out=outg[out_row[peer]])      # the column slices, in index order
            partner_work = dist.all_to_all_single(ing.view(-1), outg.view(-1), output_split_sizes=[x * plan.Dpad for x in out_split],
                                                  input_split_sizes=[x * plan.Dpad for x in in_split], group=ws.group, async_op=True)
        for first, n, st in ws.saved_blocks:        # blocks this rank evaluated in the forward: from their saved exponentials
            entry_r, operand_r = ((lib.crossclr_backward_rect_saved_xfp, ws.xf_all) if remote_xfp else
                                  (lib.crossclr_backward_rect_saved, ws.xcols))
            nat.check(entry_r(pp, _ptr(operand_r), _ptr(st), first, n, ws.temperature, ws.negative_w,
                              _ptr(ws.rz), _ptr(ws.wrz), _ptr(ws.rz_cols), _ptr(ws.wrz_cols), sw_all, _ptr(gbuf), 1, stream))
        if ws.recompute_ranges:
            ws.exchange.wait()                      # (point-to-point exchange: the slices only this recompute needs)
        for first, n in ws.recompute_ranges:        # blocks the other side of a pair evaluated: recompute
            nat.check(lib.crossclr_backward_ranks(pp, _ptr(ws.xhat), _ptr(ws.xcols), first, n, ws.temperature, ws.negative_w,
                                                  _ptr(ws.rz), _ptr(ws.wrz), _ptr(ws.rz_cols), _ptr(ws.wrz_cols), sw_all,
                                                  _ptr(gbuf), 1, stream))
        if partner_work is not None:
            _traced_wait("partner gradients", [partner_work])
            g0 = gbuf[:nel]
            for k in range(npairs):          # fixed order over the source ranks: deterministic
                g0 += ing[k]
        ws.saved_blocks = None
        ws.xf = ws.xf_all = None
    elif ws.sharded and ws.shift is None:
        if ws.partner_peers is not None:
            # Partner-gradient scheme (the default from 3 ranks): the first backward consumed the saved blocks, and the operand slices of
            # the ranks whose blocks the PARTNER evaluated were never requested (deferred exchange) -- a recomputing launch over every
            # rank's slice would read memory nobody wrote.  Every rank takes this branch together.
            raise RuntimeError("CrossCLR (sharded, partner gradients): backward was called a second time through the same graph "
                               "(retain_graph=True); the saved exponentials are released after the first backward and the late operand "
                               "slices are never exchanged in this scheme -- run the forward again, or set CROSSCLR_PARTNER_GRADS=0 on "
                               "every rank (the recomputing scheme supports repeated backwards)")
        if ws.wrz_cols is None:
            _traced_wait("statistics", [ws.stats_work])
            ws.wrz_cols = ws.rz_cols * ws.negative_w
        if ws.exchange is not None:
            ws.exchange.wait()     # this launch reads EVERY rank's slice of xcols: also the late point-to-point ones
        nat.check(lib.crossclr_backward_w(pp, _ptr(ws.xhat), _ptr(ws.xcols), world, 0, rank, ws.temperature,
                                          ws.negative_w, _ptr(ws.rz), _ptr(ws.wrz), _ptr(ws.rz_cols),
                                          _ptr(ws.wrz_cols), _sw(ws.k_rows, ws.k_cols, None), _ptr(gbuf), 1, stream))
    if grad_out.dtype == torch.float64 and grad_out.device == dev and grad_out.numel() == 1:
        go = grad_out.detach().reshape(1)
    else:
        go = grad_out.detach().to(device=dev, dtype=torch.float64).reshape(1).contiguous()
    gv = torch.empty(video.shape, dtype=video.dtype, device=dev)
    gt = torch.empty(text.shape, dtype=text.dtype, device=dev)
    with _Range("crossclr.backward_finish"):
        nat.check(lib.crossclr_backward_finish_p(pp, _ptr(gbuf), _ptr(video), _ptr(text), video.stride(0), text.stride(0),
                                                 ws.in_dtype, _ptr(ws.inv_norm), ws.temperature, _sw(None, None, ws.lw),
                                                 _ptr(go), _ptr(gv), _ptr(gt), gv.stride(0), gt.stride(0),
                                                 int(ws.prenormalized), stream))    # 0 raw rows / 1 unit rows as given / 2 unit rows, gradient w.r.t. the un-normalised vectors (projection.py)
    return gv, gt


class _device_of:
    """The C-ABI launches on the stream it is given; HIP requires that stream's device to be current (the inputs may
    live on another GPU than the process's current one, and the backward runs on an autograd engine thread)."""
    def __init__(self, t: torch.Tensor):
        # (switching devices costs ~5 us per enter/exit: only when the tensor's device is not already current)
        self._guard = torch.cuda.device(t.device) if t.is_cuda and torch.cuda.current_device() != t.device.index else None

    def __enter__(self):
        if self._guard is not None:
            self._guard.__enter__()

    def __exit__(self, *exc):
        if self._guard is not None:
            self._guard.__exit__(*exc)


def _refuse_double_backward(what: str) -> None:
    """The autograd engine runs `backward` with grad mode ON exactly when the caller asked for a graph through it
    (`create_graph=True`).  The closed-form kernels of the projection / ranking paths are not twice differentiable --
    raise instead of silently handing back a gradient that autograd would treat as a constant.  (The CrossCLR criterion itself
    carries second-order terms: `_CrossCLRGradFunction`, crossclr_second_order.)"""
    if torch.is_grad_enabled():
        raise RuntimeError(f"{what}: differentiating through the backward (create_graph=True / double backward) is not "
                           "supported by the HIP kernels; use the eager reference for second-order terms")


class _CrossCLRGradFunction(torch.autograd.Function):
    """The criterion's backward as a differentiable function of (video, text, grad_out) -- what `create_graph=True` asks for (gradient
    penalties, Hessian-vector products, meta-gradients: the reference's eager ops, trainer/loss.py:79-114, are twice differentiable).
    forward = the HIP backward of the step that is being differentiated (same kernels, same bits as the ordinary backward);
    backward = crossclr_second_order (include/crossclr.h; csrc/crossclr_kernels_hvp.h): grad_out * Hessian . u and <u, dL/d(rows)> in closed
    form on the device, exact-fp32 products whatever mode the first-order step ran in, no B x B tensor.  Single device."""

    @staticmethod
    def forward(ctx, video, text, grad_out, ws, video_c, text_c, args):
        with _device_of(video_c):
            gv, gt = _backward_impl(ws, video_c, text_c, grad_out)
        ctx.save_for_backward(video, text, grad_out)
        ctx.args = args
        return gv, gt

    @staticmethod
    def backward(ctx, u_video, u_text):
        if torch.is_grad_enabled():      # (= create_graph=True on THIS level: the caller wants to differentiate the double backward again)
            raise RuntimeError("CrossCLR_onlyIntraModality: third-order terms (create_graph=True through the double backward) are not "
                               "supported by the HIP kernels -- drop create_graph on the outer autograd call, or use the eager reference")
        video, text, grad_out = ctx.saved_tensors
        temperature, negative_w, negative_scale, loss_weight, prenormalized = ctx.args
        lib = nat.library()
        dev = video.device
        b, D = video.shape
        v, t = _row_major(video.detach()), _row_major(text.detach())
        uv = torch.zeros_like(v) if u_video is None else _row_major(u_video.detach().to(v.dtype))
        ut = torch.zeros_like(t) if u_text is None else _row_major(u_text.detach().to(t.dtype))
        plan = _plan_for(b, D, 1, 0, nat.MODE_FP32)
        k = _pack_pair(negative_scale, b, plan.bpad, dev, "negative_scale")
        lw = _pack_pair(loss_weight, b, plan.bpad, dev, "loss_weight")
        go = grad_out.detach().to(device=dev, dtype=torch.float64).reshape(1).contiguous()
        hv, ht = torch.empty_like(v), torch.empty_like(t)
        dgo = torch.empty(1, dtype=torch.float64, device=dev)
        with _device_of(v), _Range("crossclr.second_order"):
            nbytes = lib.crossclr_second_order_workspace_bytes(ctypes.byref(plan))
            work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            nat.check(lib.crossclr_second_order(ctypes.byref(plan), _ptr(v), _ptr(t), v.stride(0), t.stride(0), _IN_DTYPE[v.dtype], float(temperature),
                                                float(negative_w), _sw(k, k, lw), 1 if prenormalized else 0, _ptr(uv), _ptr(ut), uv.stride(0),
                                                ut.stride(0), _ptr(go), _ptr(work), nbytes, _ptr(hv), _ptr(ht), hv.stride(0), ht.stride(0),
                                                _ptr(dgo), _stream_for(v)))
        need = ctx.needs_input_grad
        return (hv if need[0] else None, ht if need[1] else None, dgo.reshape(grad_out.shape).to(grad_out.dtype) if need[2] else None,
                None, None, None, None)


def _second_order_grads(ctx, video, text, grad_out):
    import torch.distributed as dist
    temperature, negative_w, group, negative_scale, loss_weight, prenormalized = ctx.second_order
    if group is not None and dist.get_world_size(group) > 1:
        raise RuntimeError("CrossCLR_onlyIntraModality: create_graph=True (double backward) is not available for the row-sharded "
                           "loss; gather the features (all_gather_with_grad) and call the criterion without a process group")
    if not video.is_cuda and nat.backend() != "emu-host":
        raise RuntimeError("the HIP path needs inputs on the GPU (got a CPU tensor); there is no CPU fallback")
    video_c, text_c = ctx.row_major
    return _CrossCLRGradFunction.apply(video, text, grad_out, ctx.ws, video_c, text_c,
                                       (temperature, negative_w, negative_scale, loss_weight, prenormalized))


class _CrossCLRFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, video, text, temperature, negative_w, compute_mode, group, negative_scale, loss_weight, prenormalized):
        video_c, text_c = _row_major(video.detach()), _row_major(text.detach())
        needs_grad = any(ctx.needs_input_grad[:2])
        with _device_of(video_c):
            loss, ws = _forward_impl(video_c, text_c, temperature, negative_w, compute_mode, group, negative_scale, loss_weight,
                                     save_for_backward=needs_grad, prenormalized=prenormalized)
        if ws.step is not None and (ws.step[0].flags & nat.STEP_EAGER):
            ws.release_transient()      # the gradient product is enqueued already: backward() reads the persistent region only
        ctx.ws = ws
        ctx.second_order = (float(temperature), float(negative_w), group, negative_scale, loss_weight, bool(prenormalized))
        ctx.save_for_backward(video, text)      # (the inputs themselves: a graph through the backward needs their identity)
        ctx.row_major = (video_c, text_c)       # (aliases of the inputs unless they had to be copied into row-major form)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        video, text = ctx.saved_tensors
        if torch.is_grad_enabled() and (video.requires_grad or text.requires_grad or grad_out.requires_grad):
            # create_graph=True: the reference's eager ops (loss.py:79-114) are twice differentiable, so is this path (the same HIP
            # backward, recorded as a node whose own backward is the closed-form double backward on the device)
            gv, gt = _second_order_grads(ctx, video, text, grad_out)
            return (gv if ctx.needs_input_grad[0] else None, gt if ctx.needs_input_grad[1] else None,
                    None, None, None, None, None, None, None)
        video_c, text_c = ctx.row_major
        with _device_of(video_c):
            gv, gt = _backward_impl(ctx.ws, video_c, text_c, grad_out)
        return (gv if ctx.needs_input_grad[0] else None, gt if ctx.needs_input_grad[1] else None,
                None, None, None, None, None, None, None)


def _validate(video: torch.Tensor, text: torch.Tensor) -> None:
    # error behaviour of the reference: RuntimeError for >2-D inputs
    # (trainer/loss.py:83) and for mismatched batch sizes (trainer/loss.py:97)
    if video.dim() != 2 or text.dim() != 2:
        raise RuntimeError(f"CrossCLR expects 2-D [batch, embed_dim] inputs, got {tuple(video.shape)} and "
                           f"{tuple(text.shape)} (t() expects a tensor with <= 2 dimensions)")
    if video.shape[0] != text.shape[0]:
        raise RuntimeError(f"The size of tensor a ({text.shape[0]}) must match the size of tensor b "
                           f"({video.shape[0]}) at non-singleton dimension 1")
    if video.shape[1] != text.shape[1]:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({video.shape[0]}x{video.shape[1]} and "
                           f"{text.shape[1]}x{text.shape[0]})")
    if video.dtype != text.dtype:
        raise RuntimeError(f"expected both inputs to have the same dtype, got {video.dtype} and {text.dtype}")
    if video.dtype not in _IN_DTYPE:
        raise RuntimeError(f"unsupported input dtype {video.dtype}")
    if video.device != text.device:
        raise RuntimeError("video_features and text_features must be on the same device")


def crossclr_loss(video_features: torch.Tensor, text_features: torch.Tensor, temperature: float = 0.03,
                  negative_weight: float = 0.8, *, compute_mode: str = "auto", process_group=None,
                  negative_scale=None, loss_weight=None, prenormalized: bool = False) -> torch.Tensor:
    """Functional form of `CrossCLR_onlyIntraModality.forward` (+ the optional per-sample weights, see module doc).

    prenormalized=True: the rows ARE unit vectors already (a projection head with a fused L2-norm, or `F.normalize` upstream):
    the normalisation pass of loss.py:79-80 is skipped, and the returned gradients are those w.r.t. the unit rows as given, so
    that the upstream normalise-backward (autograd) applies."""
    _validate(video_features, text_features)
    if not video_features.is_cuda and nat.backend() != "emu-host":
        # the reference hard-codes .cuda() (trainer/loss.py:66,103,104); so does this path
        raise RuntimeError("CrossCLR HIP path needs inputs on the GPU (got a CPU tensor); there is no CPU fallback")
    if video_features.shape[0] == 0:
        return torch.full((), float("nan"), dtype=torch.float64, device=video_features.device)
    return _CrossCLRFunction.apply(video_features, text_features, float(temperature), float(negative_weight),
                                   compute_mode, process_group, negative_scale, loss_weight, bool(prenormalized))


class _AllGatherWithGrad(torch.autograd.Function):
    """all_gather whose backward returns to each rank the SUM over ranks of the gradient of its own slice."""

    @staticmethod
    def forward(ctx, x, group):
        import torch.distributed as dist
        ctx.group = group
        x = x.contiguous()
        out = torch.empty((dist.get_world_size(group) * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        return out

    @staticmethod
    def backward(ctx, grad):
        import torch.distributed as dist
        grad = grad.contiguous()
        world = dist.get_world_size(ctx.group)
        out = torch.empty((grad.shape[0] // world,) + tuple(grad.shape[1:]), dtype=grad.dtype, device=grad.device)
        dist.reduce_scatter_tensor(out, grad, op=dist.ReduceOp.SUM, group=ctx.group)
        return out, None


def all_gather_with_grad(x: torch.Tensor, process_group=None) -> torch.Tensor:
    """[b, ...] on every rank -> [world * b, ...] (rank order), differentiable: the backward reduce-scatters (sums) the
    gradient of the gathered tensor back to the rank that owns the rows.

    Two ways to train with a global batch under DistributedDataParallel (DDP averages parameter gradients over ranks):
      * gather + replicate (what open-source CLIP trainers do): `loss = criterion(all_gather_with_grad(v), all_gather_with_grad(t))`
        -- every rank evaluates the whole B x B problem (world x the work), the summed slice gradient is world x the true one
        and DDP's averaging restores it: no extra scaling;
      * sharded (preferred, world x cheaper): `criterion = CrossCLR_onlyIntraModality(..., process_group=group)` on the LOCAL
        rows -- the returned loss is the global loss, the gradients are exactly d(global loss)/d(local rows); because DDP
        averages, call `(loss * world).backward()` (or scale the learning rate) to get the gradient of the global loss.
    """
    import torch.distributed as dist
    if process_group is None:
        process_group = dist.group.WORLD
    return _AllGatherWithGrad.apply(x, process_group)


class CrossCLR_onlyIntraModality(nn.Module):
    """CrossCLR loss between two groups of embeddings -- only intra-modality alignment (ICCV 2021).

    Interface of the reference class (`trainer/loss.py:44-114`); computation by the HIP kernels.
    """

    def __init__(self, temperature=0.03, negative_weight=0.8, logger=None, *, compute_mode: str = "auto",
                 process_group=None, prenormalized: bool = False):
        super().__init__()
        # members the reference registers but never reads in forward (loss.py:52-53); kept so that
        # state_dict()/parameters()/named_children() of an existing checkpoint or optimiser match
        self.logit_scale = nn.Parameter(torch.ones([]))
        self.criterion = torch.nn.CrossEntropyLoss(reduction='none')
        self.temperature = temperature
        self.logger = logger
        self.negative_w = negative_weight
        _resolve_mode(compute_mode, 0)  # validate early
        self.compute_mode = compute_mode
        self.process_group = process_group
        self.prenormalized = prenormalized

    # cheap torch one-liners kept for API completeness (loss.py:59-66)
    def compute_loss(self, logits, mask):
        return -torch.log((torch.softmax(logits, dim=1) * mask).sum(1))

    def _get_positive_mask(self, batch_size):
        dev = self.logit_scale.device
        return 1 - torch.eye(batch_size, dtype=torch.float64, device=dev)

    def forward(self, video_features, text_features):
        """
        Inputs shape (batch, embed_dim)
        Args:
            video_features: visual embeddings (batch, embed_dim)
            text_features: text embeddings (batch, embed_dim)
        Returns: 0-dim float64 loss
        """
        # temperature / negative_w are read here, at call time, like the reference (loss.py:90-100)
        return crossclr_loss(video_features, text_features, self.temperature, self.negative_w,
                             compute_mode=self.compute_mode, process_group=self.process_group, prenormalized=self.prenormalized)

    def extra_repr(self):
        return f"temperature={self.temperature}, negative_weight={self.negative_w}, compute_mode={self.compute_mode!r}"
