"""Training path: torch.autograd Functions whose forward AND backward run in libgecco_hip.so.

The inference path fuses a whole evaluation into one C call and keeps nothing; training needs the intermediates, so
it runs the reference's op sequence unfused (models/set_transformer.py:155-168) with every op a Function:
Linear / AdaGN / GaussianActivation / pool & unpool attention (fused flash-style forward and backward kernels,
csrc/attention_bwd_f32.hip: no (B, H, N, I) tensor exists; shapes they do not take run with materialised scores on the
general strided-batched GEMM with per-operand layout flags, so no tensor is transposed for a backward product) / lift /
lower.  Cross-workgroup reductions use per-block partials summed in a fixed order: gradients are bitwise
reproducible.  EDM preconditioning and the loss are (B, N, 3) elementwise torch ops.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
from torch import Tensor

from . import _lib, hip_ops
from .hip_ops import _ptr, _stream

GN_EPS = 1e-5
_DW_BLOCKS = int(os.environ.get("GECCO_DW_BLOCKS", "768"))   # blocks a weight-gradient launch aims for (groups x output tiles)


def _f(t: Tensor) -> Tensor:
    return t.contiguous() if not t.is_contiguous() else t


def _gemm(A, B, out, *, Z, zdiv, M, N, K, lda, ldb, ldc, sA=(0, 0), sB=(0, 0), sC=(0, 0), a_km=False, b_km=False,
          scale=1.0, bias=None, a_off=0, b_off=0, c_off=0):
    lib = _lib.load()
    es = 4
    g = _lib.GeccoGemm(C.c_void_p(A.data_ptr() + a_off * es), C.c_void_p(B.data_ptr() + b_off * es),
                       _ptr(bias), C.c_void_p(out.data_ptr() + c_off * es), Z, zdiv, M, N, K, lda, ldb, ldc,
                       sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], int(a_km), int(b_km), scale)
    _lib.check(lib.gecco_gemm_f32(C.byref(g), _stream()), "gecco_gemm_f32")
    return out


def _reduce(parts: Tensor, n: int, Z: int, stride: int) -> Tensor:
    lib = _lib.load()
    out = torch.empty(n, device=parts.device, dtype=torch.float32)
    _lib.check(lib.gecco_reduce_batch_f32(_ptr(parts), _ptr(out), n, Z, stride, 0, _stream()), "gecco_reduce_batch_f32")
    return out


def _train_precision() -> str:
    """Arithmetic of the training path OUTSIDE an autocast(float16) region: the module default, except that the fp16 / mixed modes
    train in split-bf16 — without a loss scale small gradient values would flush in fp16, split-bf16 has the fp32 exponent range.
    (Under torch.autocast(float16), the reference's trainer setting, a GradScaler supplies that scale: `_lin_precision`.)"""
    p = hip_ops.default_precision()
    return "bf16x3" if p in ("fp16", "mixed", "w2") else p


def _amp_fp16() -> bool:
    """The caller runs the reference's own trainer setting — `precision="16-mixed"` of both shipped configs
    (example_configs/shapenet_airplane_unconditional.py:74, taskonomy_conditional.py:102), i.e. Lightning wraps `training_step`
    (diffusion.py:213-222) in `torch.autocast("cuda", float16)` and scales the loss with a GradScaler."""
    return (os.environ.get("GECCO_TRAIN_AMP", "1") != "0" and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.float16)


def _lin_precision() -> str:
    """Arithmetic of the training path's linears (forward, dX and dW products), decided in the FORWARD of each Function and kept
    in its ctx for the backward (which runs outside the autocast region).  Under `torch.autocast(float16)` the reference's
    nn.Linear / in_proj products run with fp16 operands and fp32 accumulation, and so do ours then ("fp16": one MFMA per product
    instead of split-bf16's three; fp32 parameters, fp32 tensors between the kernels, fp32 weight gradients — torch rounds those
    to fp16 too); the caller's GradScaler keeps the gradients inside fp16's range exactly as it does for the reference (an
    overflow becomes inf in the weight gradients, which the scaler detects and skips).  Without autocast: `_train_precision()`."""
    return "fp16" if _amp_fp16() else _train_precision()


def _new(*shape, like: Tensor) -> Tensor:
    return torch.empty(*shape, device=like.device, dtype=torch.float32)


# ------------------------------------------------------------------------------------------- weight images
class WeightImages:
    """The split-bf16 weight images of a training step, made AHEAD in batched launches.

    Every linear of the split-bf16 path streams a tiled bf16 hi | lo image of its weight (csrc/gemm_f32_dma.hip); made per
    call that is one small launch per linear in the forward and, in the backward, a transposed copy of W plus the image of
    that copy for the dX product: ~260 launches of ~5 us per step for the shipped model.  The first step RECORDS which
    (weight view, orientation) pairs its linears asked for; from then on `prepare()` — called by `Diffusion.training_step`
    — writes all of them with `gecco_split_bf16_images_f32` (<= 96 weights per launch; the W^T images straight from W) and
    the linears only look their image up.  An image is valid for the weight values `prepare()` saw: the key carries the
    tensor's version counter (in-place torch updates miss), and `FusedAdamEMA.step()` — which updates through raw pointers
    — calls `invalidate()`.  A miss falls back to the per-call image, always correct."""

    def __init__(self):
        self.plan: dict[tuple, tuple] = {}     # key -> (tensors kept alive, job descriptions)
        self.images: dict[tuple, Tensor] = {}  # key -> image (a view of self.pool), valid while self.armed
        self.versions: dict[tuple, tuple] = {}
        self.pool: Tensor | None = None
        self.armed = False
        self.fresh = False                      # prepare() ran and no grad-enabled forward has used its images yet
        self.recording = False
        self.step = 0                           # prepare() calls so far
        self.used: dict[tuple, int] = {}        # key -> the step it was last asked for (stale keys are dropped)

    @staticmethod
    def _key(kind: str, *ws: Tensor) -> tuple:
        return (kind,) + tuple((w.data_ptr(), tuple(w.shape), tuple(w.stride())) for w in ws)

    def lookup(self, kind: str, *ws: Tensor, prec: str | None = None) -> Tensor | None:
        """The ready image for this weight (pair), or None.  kind: "n" image of W, "t" image of W^T, "pair" W1 | W2.
        prec: "bf16x3" (default: the training precision) or "fp16" — the fp16 images of the autocast(float16) setting are
        their own entries (kind + "16")."""
        prec = _train_precision() if prec is None else prec
        if prec not in ("bf16x3", "fp16", "a16", "h8"):
            return None
        if prec == "fp16":
            if any(w.shape[-1] % 32 for w in ws) or (kind == "t" and ws[0].shape[0] % 32):
                return None
            kind = kind + "16"
        elif prec == "a16":   # the A-stationary fp16 kernels' streams (gecco_astat16_images_f32): 64-column tiles, 64-k groups
            if any(w.shape[0] % 64 or w.shape[1] % 64 for w in ws):
                return None
            kind = kind + "a16"
        elif prec == "h8":    # the h8 kernel's streams (gecco_h8_images_f32): fp16 main part + fp8 correction, 64-column tiles
            if kind == "t" or any(w.shape[0] % 64 or w.shape[1] % 64 for w in ws):
                return None
            kind = kind + "h8"
        key = self._key(kind, *ws)
        self.used[key] = self.step
        if self.armed:
            img = self.images.get(key)
            if img is not None and self.versions[key] == tuple(w._version for w in ws):
                return img
        if self.recording and key not in self.plan and all(w.stride(-1) == 1 for w in ws):
            self.plan[key] = tuple(ws)
        return None

    def invalidate(self) -> None:
        self.armed = False

    def begin_forward(self) -> None:
        """Called at the start of every grad-enabled `Diffusion.forward`.  The images are valid for the weight values
        `prepare()` saw, and writes through `param.data` (EMA weight swaps, hand-written updates) do not move the version
        counter the lookups compare — so the images serve exactly ONE forward (and its backward): the first one after
        `prepare()`.  Any other grad-enabled forward (a `model.loss(...)` outside `training_step`, a second model, a validation
        loss computed with grad) disarms them and its linears make their images per call — always correct."""
        if not self.fresh:
            self.armed = False
        self.fresh = False

    @torch.no_grad()
    def prepare(self) -> None:
        """(Re)build every recorded image from the current weight values; arms the lookups."""
        if os.environ.get("GECCO_WEIGHT_IMAGES", "1") == "0":   # per-call images only (A/B runs, tests)
            self.armed = self.recording = False
            return
        self.recording = True
        self.step += 1
        # weights nobody asked for during the last two steps (a model that is gone, a branch no longer taken) leave the plan
        stale = [k for k in self.plan if self.used.get(k, 0) < self.step - 2]
        for k in stale:
            self.plan.pop(k, None)
            self.images.pop(k, None)
            self.versions.pop(k, None)
            self.used.pop(k, None)
        if not self.plan:
            self.armed = False
            return
        lib = _lib.load()
        jobs, offs, total = [], {}, 0
        for key, ws in self.plan.items():
            kind = key[0]
            fmt = 3 if kind.endswith("h8") else 2 if kind.endswith("a16") else 1 if kind.endswith("16") else 0
            tr = kind.startswith("t")
            nbytes = 0
            for w in ws:
                nout, k = (w.shape[1], w.shape[0]) if tr else (w.shape[0], w.shape[1])
                jobs.append((key, w, nout, k, tr, total + nbytes, fmt))
                nbytes += (lib.gecco_split_bf16_image_bytes, lib.gecco_split_f16_image_bytes, lib.gecco_astat16_image_bytes,
                           lib.gecco_h8_image_bytes)[fmt](nout, k)
            offs[key] = (total, nbytes)
            total += (nbytes + 255) // 256 * 256
        dev = jobs[0][1].device
        if self.pool is None or self.pool.numel() < total or self.pool.device != dev:
            self.pool = torch.empty(total, dtype=torch.uint8, device=dev)
        base = self.pool.data_ptr()
        for fmt, fn, name in ((0, lib.gecco_split_bf16_images_f32, "gecco_split_bf16_images_f32"),
                              (1, lib.gecco_split_f16_images_f32, "gecco_split_f16_images_f32"),
                              (2, lib.gecco_astat16_images_f32, "gecco_astat16_images_f32"),
                              (3, lib.gecco_h8_images_f32, "gecco_h8_images_f32")):
            sel = [j for j in jobs if j[6] == fmt]
            if not sel:
                continue
            arr = (_lib.GeccoSplitJob * len(sel))()
            for j, (key, w, nout, k, tr, off, _) in enumerate(sel):
                arr[j] = _lib.GeccoSplitJob(w.data_ptr(), base + off, nout, k, w.stride(0), int(tr))
            _lib.check(fn(arr, len(sel), _stream()), name)
        self.images = {key: self.pool[o:o + n] for key, (o, n) in offs.items()}
        self.versions = {key: tuple(w._version for w in ws) for key, ws in self.plan.items()}
        self.armed = True
        self.fresh = True


WEIGHT_IMAGES = WeightImages()


def _resolve(prec: str | None, rows: int, K: int, Nout: int) -> str:
    """The arithmetic of ONE product: `prec` (None: the training precision), except that an "fp16" request for a shape the fp16
    LDS-DMA kernel does not take goes back to the training precision (hip_ops.linear would run it on the exact-fp32 kernel)."""
    if prec is None:
        return _train_precision()
    if prec == "fp16" and not _lib.load().gecco_linear_image_ok_f16(rows, K, Nout, 0):
        return _train_precision()
    return prec


def _image_ok(rows: int, K: int, Nout: int, prec: str = "bf16x3") -> bool:
    lib = _lib.load()
    return bool(lib.gecco_linear_image_ok_f16(rows, K, Nout, 0) if prec == "fp16" else lib.gecco_linear_image_ok(rows, K, Nout, 0))


# ------------------------------------------------------------------------------------------- Linear
def _linear_dx_dot(dy: Tensor, W: Tensor, x: Tensor, prec: str | None = None, residual: Tensor | None = None):
    """(dx, gst): dx = dy W and the partial sums {sum dx, sum dx * x} per (sample, row tile, column) that the AdaGN backward of the
    tensor x needs (`_adagn_backward(..., gst=)`), from the GEMM's epilogue (`gecco_linear_dotstats_f32`) instead of a
    `col_dot_stats` pass over dx and x; gst None where the LDS-DMA kernels do not take the shape (the caller's AdaGN backward then
    runs that pass)."""
    lib = _lib.load()
    B, R, Nout = dy.shape
    K = W.shape[1]
    if dy.dtype == torch.float16:   # du stored as halves (`_du16_ok` checked the shape): the fp16 kernel reads its tiles as they are
        dx = _new(B, R, K, like=x)
        gst = _new(B, lib.gecco_linear_row_tiles(R), 2, K, like=x)
        img = WEIGHT_IMAGES.lookup("t", W, prec="fp16")
        Wt, ws = (None, img) if img is not None else (W.t().contiguous(), hip_ops._ws((K + 127) // 128 * 128 * Nout * 4, dy.device))
        _lib.check(lib.gecco_linear_dotstats_a16_f32(hip_ops._ptr16(dy), _ptr(Wt), _ptr(x), _ptr(residual), _ptr(dx), _ptr(gst), B, R, Nout, K,
                                                     C.c_void_p(ws.data_ptr()), _stream()), "gecco_linear_dotstats_a16_f32")
        return dx, gst
    assert residual is None, "a residual with the partials: fp16 dy only"
    prec = _resolve(prec, R, Nout, K)
    if (os.environ.get("GECCO_TRAIN_DOTSTATS", "1") == "0" or prec not in ("fp32", "bf16x3", "fp16")
            or not lib.gecco_linear_actbwd_ok(R, Nout, K, hip_ops.PRECISIONS[prec])):
        return _linear_dx(dy, W, prec=prec), None
    dx = _new(B, R, K, like=dy)
    gst = _new(B, lib.gecco_linear_row_tiles(R), 2, K, like=dy)
    img = WEIGHT_IMAGES.lookup("t", W, prec=prec) if prec in ("bf16x3", "fp16") else None
    if img is not None:
        Wt, ws = None, img
    else:
        Wt = W.t().contiguous()
        ws = hip_ops._ws((K + 127) // 128 * 128 * Nout * 4, dy.device) if prec != "fp32" else None
    _lib.check(lib.gecco_linear_dotstats_f32(_ptr(dy), _ptr(Wt), _ptr(x), _ptr(dx), _ptr(gst), B, R, Nout, K, hip_ops.PRECISIONS[prec],
                                             C.c_void_p(ws.data_ptr()) if ws is not None else None, _stream()), "gecco_linear_dotstats_f32")
    return dx, gst


def _linear_dx(dy: Tensor, W: Tensor, residual: Tensor | None = None, prec: str | None = None, out_f16: bool = False) -> Tensor:
    """dx = dy W (+ residual: another gradient contribution to the same tensor, added in the GEMM's epilogue).
    prec: the arithmetic the Function's forward chose (`_lin_precision()`); None: the training precision.
    fp16 tensors (the `_io16_ok` layers under autocast(float16)): dy may be an fp16 tensor (its tiles go to LDS as they are), and with
    out_f16 the result leaves as one (no residual then) — the fp16 LDS-DMA kernel either way."""
    B, R, Nout = dy.shape
    K = W.shape[1]
    if out_f16 and residual is None and dy.dtype == torch.float32 and _a16_ok("fp16", R, Nout, K) and K >= 128:
        # fp32 gradient in, fp16 gradient out (out_proj's dX in an `_io16_ok` layer): the A-stationary kernel's fp16 row-major epilogue —
        # dy is read once into registers, against the LDS-DMA kernel's fp32 A tiles (141 -> ~60 us at the shipped shape)
        ws, ready = _a16_stream("t", W, dev=dy.device)
        return hip_ops.linear_kvq_f16(dy, None, W.t() if ready else W.t().contiguous(), None, head_dim=0, wsplit=ws, image_ready=ready)
    if dy.dtype == torch.float16 or out_f16:
        img = WEIGHT_IMAGES.lookup("t", W, prec="fp16")
        return hip_ops.linear_f16io(dy, None if img is not None else W.t().contiguous(), residual=residual, out_f16=out_f16, w_image=img,
                                    w_shape=(K, Nout))
    prec = _resolve(prec, R, Nout, K)
    # (with a residual the A-stationary form measured the same as the LDS-DMA one, 19.31 vs 19.27 ms per step: only the plain product)
    if residual is None and _a16_ok(prec, R, Nout, K):
        dx = _new(B, R, K, like=dy)
        ws, ready = _a16_stream("t", W, dev=dy.device)
        _lib.check(_lib.load().gecco_linear_astat16_f32(_ptr(dy), None, None, None if ready else _ptr(_f(W)), None, K, _ptr(dx), None, None, 0, None,
                                                        _ptr(residual), 1, B, R, Nout, C.c_void_p(ws.data_ptr()), _stream()),
                   "gecco_linear_astat16_f32")
        return dx
    if R >= 64 and Nout % 16 == 0 and K % 4 == 0:
        # linear(dy, W^T): the fused LDS-DMA GEMM; the image of W^T comes ready from the step's batched launch when it is
        # there (WeightImages), else from a transposed copy of the (small) weight
        img = WEIGHT_IMAGES.lookup("t", W, prec=prec) if _image_ok(R, Nout, K, prec) else None
        if img is not None:
            return hip_ops.linear(dy, None, residual=residual, precision=prec, w_image=img, w_shape=(K, Nout))
        return hip_ops.linear(dy, W.t().contiguous(), residual=residual, precision=prec)
    dx = _gemm(dy, W, _new(B, R, K, like=dy), Z=1, zdiv=1, M=B * R, N=K, K=Nout, lda=Nout, ldb=K, ldc=K, b_km=True)  # W read k-major
    return dx if residual is None else dx + residual


# Weight gradients on a SIDE stream.  The dX products form the backward's critical path (each feeds the next backward op); the
# weight gradients feed nothing until the optimizer.  Both families of kernels keep the matrix pipe 28 - 40 % busy on their own
# (profiles/r02za_train_pmc_summary.txt), so the dW kernel of a linear is issued on a second stream and shares the CUs with the
# dX chain running ahead on the main one.  Ordering: the side stream waits for the main stream's work issued so far (dy and x
# exist), the caching allocator is told about the cross-stream uses (record_stream), and ONE engine callback per backward pass
# makes the main stream wait for the side stream when the pass ends — after loss.backward() returns, every gradient is
# ordered on the main stream as before.  Mid-pass consumers on the main stream (a gradient that is ACCUMULATED into an
# existing .grad, the data-parallel reducer's bucket launch) synchronise explicitly.
_SIDE = {"stream": None, "pending": False}


def _side_enabled() -> bool:
    # (inside a hipGraph capture the fork / join is captured with it — one level of fork, as the two-stream forward's: opt-in,
    # GECCO_TRAIN_DW_STREAM_CAPTURE=1, tools/debug/graph_train.py)
    return os.environ.get("GECCO_TRAIN_DW_STREAM", "1") != "0" and torch.cuda.is_available() and \
        (not torch.cuda.is_current_stream_capturing() or os.environ.get("GECCO_TRAIN_DW_STREAM_CAPTURE", "0") == "1")


def sync_side_stream() -> None:
    """The main (current) stream waits for the weight-gradient work issued on the side stream so far."""
    if _SIDE["pending"] and _SIDE["stream"] is not None:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])
        _SIDE["pending"] = False


def _linear_dw(dy: Tensor, x: Tensor, want_db: bool = False, pro=None, leaf: Tensor | None = None, prec: str | None = None,
               side_ok: bool = False):
    """leaf: the parameter this gradient is FOR, when the caller knows that nothing will read the result before the pass ends —
    a leaf that is not a view (a view's gradient is scattered into its base by autograd, on the main stream, right away) and
    has no .grad yet (autograd then keeps the tensor instead of adding it into an existing one).  Only then the side stream.
    side_ok (round 6): the weight is a third of nn.MultiheadAttention's packed in_proj handed out by `InProjSplitFn`, whose backward joins
    the thirds' gradients ON the side stream (`_inproj_side_ok` checked the packed parameter at forward time)."""
    if not _side_enabled() or not (side_ok or (leaf is not None and leaf.is_leaf and leaf._base is None and leaf.grad is None)):
        return _linear_dw_main(dy, x, want_db, pro, prec)
    if _SIDE["stream"] is None:
        _SIDE["stream"] = torch.cuda.Stream()
    side, main = _SIDE["stream"], torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        res = _linear_dw_main(dy, x, want_db, pro, prec)
    for t in (dy, x) + (tuple(pro) if pro is not None else ()):
        t.record_stream(side)
    for t in (res if isinstance(res, tuple) else (res,)):
        t.record_stream(main)
    _SIDE["pending"] = True
    try:   # one callback per call (the first to run does the wait, the rest find nothing pending): no state to go stale if a
        # backward pass is abandoned half way
        torch.autograd.Variable._execution_engine.queue_callback(sync_side_stream)
    except RuntimeError:   # not inside a backward pass: order it right away
        sync_side_stream()
    return res


_TN_CTR: dict = {}


def _tn_counters(dev, tiles: int) -> Tensor:
    """`tiles` zeroed counters for one weight-gradient launch (gecco_gemm_tn_f16_ex_f32 leaves them zero again): slots of one ring per
    device, handed out round-robin — a slot comes round again thousands of launches later, long after its launch has finished."""
    key = (dev.type, dev.index)
    st = _TN_CTR.get(key)
    if st is None:
        st = _TN_CTR[key] = [torch.zeros(1 << 16, dtype=torch.int32, device=dev), 0]
        torch.cuda.synchronize(dev)   # (once: the fill is ordered in front of every stream that will use the ring)
    if st[1] + tiles > st[0].numel():
        st[1] = 0
    off = st[1]
    st[1] += tiles
    return st[0][off:off + tiles]


def _linear_dw_main(dy: Tensor, x: Tensor, want_db: bool = False, pro=None, prec: str | None = None):
    """dW = dy^T x: both operands read k-major (contraction over their rows); partials summed in a fixed order.
    want_db: also the bias gradient db = column sums of dy -> (dW, db); the split-bf16 kernel forms it from the dy tiles
    it stages anyway.  pro = (a, o): the linear's input was AdaGN(x) = a x + o (per sample and column) — the split-bf16 kernel
    applies it while it stages x; elsewhere the normalised tensor is formed first.  prec "fp16": the same kernel with one fp16
    plane per operand (`gecco_gemm_tn_f16_f32`)."""
    B, R, K = x.shape
    Nout = dy.shape[2]
    prec = _train_precision() if prec is None else prec
    assert dy.dtype == torch.float32 or (prec == "fp16" and R % 32 == 0 and Nout % 8 == 0 and K % 4 == 0), "fp16 gradients exist only on the fp16 path"
    tn_ok = prec in ("bf16x3", "fp16") and R % 32 == 0 and Nout % 4 == 0 and K % 4 == 0
    if pro is not None and not tn_ok:
        x, pro = hip_ops.affine_apply(x, pro[0], pro[1]), None
    if tn_ok and prec == "fp16":
        if B == 1 and R >= 4096:   # one long row block: groups of rows (see below)
            cap = max(64, 768 // (-(-Nout // 128) * -(-K // 128)))
            g = max((d for d in range(1, cap + 1) if R % (32 * d) == 0), default=1)
            if g > 1 and pro is None:
                dy, x = dy.view(g, R // g, Nout), x.view(g, R // g, K)
                B, R = g, R // g
        tiles = _lib.load().gecco_gemm_tn_f16_tiles(Nout, K)
        G = min(B, max(1, -(-_DW_BLOCKS // tiles)))
        group = -(-B // G)
        G = -(-B // group)
        parts = _new(G, Nout, K, like=x)
        cparts = _new(G, Nout, like=x) if want_db else None
        if os.environ.get("GECCO_TRAIN_DW_REDUCE", "launch") == "kernel":
            # OPT-IN (measured and lost, profiles/r06_negative_results.txt): the fixed-order sum of the group partials inside the launch —
            # the last block to finish a tile adds them in group order, the bits `_reduce` gives without its ~100 launches per step.  One
            # block per tile then reads 24 x 64 KiB behind everybody else (18 CUs busy, 238 idle): 16.75 -> 21.4 ms per step; the separate
            # reduction spreads the same bytes over the whole chip in ~10 us
            dW = _new(Nout, K, like=x)
            db = _new(Nout, like=x) if want_db else None
            ctr = _tn_counters(x.device, tiles)
            a16, b16 = dy.dtype == torch.float16, x.dtype == torch.float16
            assert not (b16 and pro is not None)
            _lib.check(_lib.load().gecco_gemm_tn_f16_ex_f32(
                C.c_void_p(dy.data_ptr()), int(a16), C.c_void_p(x.data_ptr()), int(b16), _ptr(pro[0]) if pro is not None else None,
                _ptr(pro[1]) if pro is not None else None, _ptr(parts), _ptr(cparts), _ptr(dW), _ptr(db), C.c_void_p(ctr.data_ptr()),
                B, R, Nout, K, group, _stream()), "gecco_gemm_tn_f16_ex_f32")
            return (dW, db) if want_db else dW
        if dy.dtype == torch.float16 and x.dtype == torch.float16:   # both operands fp16 tensors (`_y16_ok`): slabs by DMA
            assert pro is None
            _lib.check(_lib.load().gecco_gemm_tn_f16_ex_f32(C.c_void_p(dy.data_ptr()), 1, C.c_void_p(x.data_ptr()), 1, None, None, _ptr(parts),
                                                            _ptr(cparts), None, None, None, B, R, Nout, K, group, _stream()),
                       "gecco_gemm_tn_f16_ex_f32")
        elif dy.dtype == torch.float16:   # du of an MLP's backward stored as halves (`_du16_ok`)
            assert x.dtype == torch.float32
            _lib.check(_lib.load().gecco_gemm_tn_f16_a16_f32(hip_ops._ptr16(dy), _ptr(x), _ptr(pro[0]) if pro is not None else None,
                                                             _ptr(pro[1]) if pro is not None else None, _ptr(parts), _ptr(cparts), B, R,
                                                             Nout, K, group, _stream()), "gecco_gemm_tn_f16_a16_f32")
        elif x.dtype == torch.float16:   # the fp16 hidden layer of an MLP (_keep_h16): its tiles go to LDS as they are
            assert pro is None
            _lib.check(_lib.load().gecco_gemm_tn_f16_b16_f32(_ptr(dy), hip_ops._ptr16(x), _ptr(parts), _ptr(cparts), B, R, Nout, K, group,
                                                             _stream()), "gecco_gemm_tn_f16_b16_f32")
        else:
            _lib.check(_lib.load().gecco_gemm_tn_f16_f32(_ptr(dy), _ptr(x), _ptr(pro[0]) if pro is not None else None,
                                                         _ptr(pro[1]) if pro is not None else None, _ptr(parts), _ptr(cparts), B, R,
                                                         Nout, K, group, _stream()), "gecco_gemm_tn_f16_f32")
        dW = _reduce(parts, Nout * K, G, Nout * K).reshape(Nout, K)
        return (dW, _reduce(cparts, Nout, G, Nout)) if want_db else dW
    if pro is not None:
        tiles = -(-Nout // 128) * -(-K // 128)
        G = min(B, max(1, -(-_DW_BLOCKS // tiles)))
        group = -(-B // G)
        G = -(-B // group)
        parts = _new(G, Nout, K, like=x)
        cparts = _new(G, Nout, like=x) if want_db else None
        _lib.check(_lib.load().gecco_gemm_tn_x3_pro_f32(_ptr(dy), _ptr(x), _ptr(pro[0]), _ptr(pro[1]), _ptr(parts), _ptr(cparts), B, R,
                                                        Nout, K, group, _stream()), "gecco_gemm_tn_x3_pro_f32")
        dW = _reduce(parts, Nout * K, G, Nout * K).reshape(Nout, K)
        return (dW, _reduce(cparts, Nout, G, Nout)) if want_db else dW
    if B == 1 and R >= 4096:
        # one long row block (the conditioner's texel matrix): cut it into groups for the per-group partials below — enough of
        # them that groups x output tiles fill the chip (a 96 x 384 gradient has three tiles)
        cap = max(64, 768 // (-(-Nout // 128) * -(-K // 128)))
        g = max((d for d in range(1, cap + 1) if R % (32 * d) == 0), default=1)
        if g > 1:
            dy, x = dy.view(g, R // g, Nout), x.view(g, R // g, K)
            B, R = g, R // g
    if want_db:
        if not (prec == "bf16x3" and R % 32 == 0 and Nout % 4 == 0 and K % 4 == 0):
            return _linear_dw_main(dy, x, prec=prec), _linear_db(dy)
        tiles = -(-Nout // 128) * -(-K // 128)
        G = min(B, max(1, -(-_DW_BLOCKS // tiles)))
        group = -(-B // G)
        G = -(-B // group)
        parts, cparts = _new(G, Nout, K, like=x), _new(G, Nout, like=x)
        _lib.check(_lib.load().gecco_gemm_tn_x3_bias_f32(_ptr(dy), _ptr(x), _ptr(parts), _ptr(cparts), B, R, Nout, K, group,
                                                         _stream()), "gecco_gemm_tn_x3_bias_f32")
        return _reduce(parts, Nout * K, G, Nout * K).reshape(Nout, K), _reduce(cparts, Nout, G, Nout)
    if prec == "bf16x3" and R % 32 == 0 and Nout % 4 == 0 and K % 4 == 0:
        # split-bf16 MFMA with transposed LDS reads (gemm_tn_x3.hip); one partial per group of samples, groups
        # sized so that ~1000 blocks fill the chip
        tiles = -(-Nout // 128) * -(-K // 128)
        G = min(B, max(1, -(-_DW_BLOCKS // tiles)))
        group = -(-B // G)
        G = -(-B // group)
        parts = _new(G, Nout, K, like=x)
        _lib.check(_lib.load().gecco_gemm_tn_x3_f32(_ptr(dy), _ptr(x), _ptr(parts), B, R, Nout, K, group, _stream()),
                   "gecco_gemm_tn_x3_f32")
        return _reduce(parts, Nout * K, G, Nout * K).reshape(Nout, K)
    parts = _gemm(dy, x, _new(B, Nout, K, like=x), Z=B, zdiv=1, M=Nout, N=K, K=R, lda=Nout, ldb=K, ldc=K,
                  sA=(R * Nout, 0), sB=(R * K, 0), sC=(Nout * K, 0), a_km=True, b_km=True)
    return _reduce(parts, Nout * K, B, Nout * K).reshape(Nout, K)


def _linear_db(dy: Tensor) -> Tensor:
    st = hip_ops.col_stats(dy)  # (B, T, 2, Nout): [..., 0, :] = column sums
    Nout = dy.shape[2]
    return _reduce(st, Nout, st.shape[0] * st.shape[1], 2 * Nout)


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b (+ residual) on (B, R, K).  `residual` is the skip connection of the reference's
    `x = x + f(...)` (models/set_transformer.py:164-166) folded into the GEMM's epilogue; its gradient is dy itself."""

    @staticmethod
    def forward(ctx, x, W, b, residual=None, want_stats=False):
        """want_stats: also return the GroupNorm partial sums of the output, (B, T, 2, Nout), from the GEMM's epilogue — the
        next AdaGN takes them instead of a pass over the tensor (not differentiable: AdaGNFn's backward owns that path)."""
        x = _f(x)
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        ctx.side_ok = bool(getattr(W, "_gecco_side_ok", False))
        prec = ctx.prec = _lin_precision()
        res = None if residual is None else _f(residual)
        if x.dtype == torch.float16:   # an fp16 tensor of an `_io16_ok` layer (the unpool attention's output): fp16 A tiles, fp32 result
            img = WEIGHT_IMAGES.lookup("n", W, prec="fp16")
            out = hip_ops.linear_f16io(x, None if img is not None else W, b, residual=res, want_stats=want_stats, w_image=img,
                                       w_shape=tuple(W.shape))
            if want_stats:
                ctx.mark_non_differentiable(out[1])
                ctx.set_materialize_grads(False)
            return out
        img = WEIGHT_IMAGES.lookup("n", W, prec=prec) if _image_ok(x.shape[1], W.shape[1], W.shape[0], prec) else None
        kw = dict(precision=prec, w_image=img, w_shape=tuple(W.shape)) if img is not None else dict(precision=prec)
        if want_stats:
            y, st = hip_ops.linear(x, None if img is not None else W, b, residual=res, want_stats=True, **kw)
            ctx.mark_non_differentiable(st)
            ctx.set_materialize_grads(False)   # else backward is handed a zero-filled (B, T, 2, Nout) tensor for st: one fill launch per call
            return y, st
        return hip_ops.linear(x, None if img is not None else W, b, residual=res, **kw)

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        if dy is None:   # (materialize off) only the statistics were used: they carry no gradient
            return (None,) * len(ctx.needs_input_grad)
        x, W = ctx.saved_tensors
        dy = _f(dy)
        prec = ctx.prec
        # (an fp16 input's gradient leaves as an fp16 tensor: its consumer — the attention backward — reads it as an fp16 operand)
        dx = _linear_dx(dy, W, prec=prec, out_f16=x.dtype == torch.float16) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.has_bias and ctx.needs_input_grad[2] and ctx.needs_input_grad[1]:
            dW, db = _linear_dw(dy, x, want_db=True, leaf=W, prec=prec, side_ok=ctx.side_ok)
        elif ctx.needs_input_grad[1]:
            dW = _linear_dw(dy, x, leaf=W, prec=prec, side_ok=ctx.side_ok)
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = _linear_db(dy)
        n = len(ctx.needs_input_grad)        # 3 .. 5: called without / with a residual (and the statistics flag)
        dres = dy if n > 3 and ctx.needs_input_grad[3] else None
        return (dx, dW, db, dres, None)[:n]


class LinearPairFn(torch.autograd.Function):
    """(x @ W1^T + b1, x @ W2^T + b2): the two projections of `broadcast_norm(x)` (AttentionPool.kv_proj,
    models/set_transformer.py:49, and the query third of nn.MultiheadAttention's in_proj, :112) in one launch that reads x
    once; in the backward the second dx product adds onto the first in its epilogue (no separate accumulation pass)."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2):
        x = _f(x)
        ctx.save_for_backward(x, W1, W2)
        ctx.bias = (b1 is not None, b2 is not None)
        prec = ctx.prec = _lin_precision()
        img = None
        if W1.shape[0] % 128 == 0 and W2.shape[0] % 128 == 0 and _image_ok(x.shape[1], W1.shape[1], W1.shape[0] + W2.shape[0], prec):
            img = WEIGHT_IMAGES.lookup("pair", W1, W2, prec=prec)
        if img is not None:
            return hip_ops.linear_pair(x, W1, b1, W2, b2, precision=prec, w_image=img)
        return hip_ops.linear_pair(x, W1, b1, _f(W2), b2, precision=prec)

    @staticmethod
    def backward(ctx, d1, d2):
        x, W1, W2 = ctx.saved_tensors
        d1, d2 = _f(d1), _f(d2)
        need = ctx.needs_input_grad
        prec = ctx.prec
        dx = _linear_dx(d2, W2, residual=_linear_dx(d1, W1, prec=prec), prec=prec) if need[0] else None
        out = [dx]
        for d, has_b, iw, Wl in ((d1, ctx.bias[0], 1, W1), (d2, ctx.bias[1], 3, W2)):
            if need[iw] and has_b and need[iw + 1]:
                out += list(_linear_dw(d, x, want_db=True, leaf=Wl, prec=prec))
            else:
                out += [_linear_dw(d, x, leaf=Wl, prec=prec) if need[iw] else None, _linear_db(d) if has_b and need[iw + 1] else None]
        return tuple(out)


# ------------------------------------------------------------------------------------------- AdaGN / GroupNorm
class AdaGNFn(torch.autograd.Function):
    """y = scale(t) * GroupNorm(x) + bias(t) on (B, R, C); params None -> plain GroupNorm."""

    @staticmethod
    def forward(ctx, x, t, sw, sb, bw, bb, G, eps, passthrough=False, stats=None):
        """passthrough: also return x itself, for the residual connection around the normalised branch — the gradient
        that comes back through it is added inside this Function's backward kernel instead of by an autograd pass.
        stats: the partial sums {sum x, sum x^2} per (sample, row tile, channel) when the producer of x already formed them
        (LinearFn(..., want_stats=True)); otherwise one pass over x here."""
        x = _f(x)
        ctx.set_materialize_grads(False)
        ctx.passthrough = passthrough
        a, o, stats, t2 = _adagn_coeffs(x, t, sw, sb, bw, bb, G, eps, stats)
        ctx.save_for_backward(x, stats, t2, sw, sb, bw)
        ctx.G, ctx.eps, ctx.affine = G, eps, sw is not None
        ctx.t_shape = None if t is None else tuple(t.shape)
        y = hip_ops.affine_apply(x, a, o)
        return (y, x) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, stats, t2, sw, sb, bw = ctx.saved_tensors
        none = (None,) * 9
        if dy is None:   # only the skip connection carried a gradient
            return (dskip, *none)
        want_dt = ctx.affine and ctx.needs_input_grad[1]
        dx, dsw, dsb, dbw, dbb, dt = _adagn_backward(x, stats, t2, sw, sb, _f(dy), dskip, ctx.G, ctx.eps, ctx.affine,
                                                     bw=bw if want_dt else None)
        if not ctx.affine:
            return (dx, *none)
        return dx, (dt.reshape(ctx.t_shape) if dt is not None else None), dsw, dsb, dbw, dbb, None, None, None, None


def _adagn_backward(x, stats, t2, sw, sb, dy, dskip, G, eps, affine, gst=None, bw=None):
    """Backward of y = scale(t) GroupNorm(x) + bias(t) given dy (and the gradient `dskip` that reached x through a skip
    connection, added in the same pass): dx and the gradients of the scale / bias linears, and — with `bw` given: a caller
    differentiates with respect to the noise level — of the embedding t: dt = ds scale_w + dz bias_w.
    gst: the {sum dy, sum dy x} partials when the GEMM that produced dy already formed them (`_linear_dx_dot`)."""
    lib = _lib.load()
    B, R, Cc = x.shape
    if gst is None:
        # {sum dy, sum dy x} partials in col_dot_stats' own row tiling (the forward statistics may come from a GEMM epilogue
        # with another one)
        gst = _new(B, lib.gecco_stats_row_tiles(R), 2, Cc, like=x)
        _lib.check(lib.gecco_col_dot_stats_f32(_ptr(dy), _ptr(x), _ptr(gst), B, R, Cc, _stream()), "col_dot_stats")
    cA, cB, cC, ds, dz = (_new(B, Cc, like=x) for _ in range(5))
    p = _lib.GeccoAdaGN(_ptr(sw), _ptr(sb), None, None) if affine else None
    ctxd = 0 if t2 is None else t2.shape[1]
    _lib.check(lib.gecco_adagn_bwd_coeffs_f32(_ptr(stats), stats.shape[1], _ptr(gst), gst.shape[1], R, _ptr(t2), ctxd,
                                              C.byref(p) if p is not None else None, _ptr(cA), _ptr(cB), _ptr(cC),
                                              _ptr(ds), _ptr(dz), B, Cc, G, eps, _stream()), "adagn_bwd_coeffs")
    dx = torch.empty_like(x)
    _lib.check(lib.gecco_affine2_apply_add_f32(_ptr(dy), _ptr(x), _ptr(cA), _ptr(cB), _ptr(cC),
                                               None if dskip is None else _ptr(_f(dskip)), _ptr(dx), B, R, Cc, _stream()),
               "affine2_apply_add")
    if not affine:
        return dx, None, None, None, None, None
    dsw, dbw = _new(Cc, ctxd, like=x), _new(Cc, ctxd, like=x)
    dsb, dbb = _new(Cc, like=x), _new(Cc, like=x)
    _lib.check(lib.gecco_adagn_param_grads_f32(_ptr(ds), _ptr(dz), _ptr(t2), B, Cc, ctxd, _ptr(dsw), _ptr(dsb),
                                               _ptr(dbw), _ptr(dbb), _stream()), "adagn_param_grads")
    # (B, C) x (C, ctx): a few thousand multiply-adds, formed only for a caller that wants the noise level's gradient
    dt = (ds @ sw.float() + dz @ bw.float()) if bw is not None else None
    return dx, dsw, dsb, dbw, dbb, dt


def _adagn_coeffs(x, t, sw, sb, bw, bb, G, eps, stats):
    """(a, o, stats, t2) with AdaGN(x) = a x + o: the forward half every AdaGN Function shares."""
    B, R, _ = x.shape
    stats = hip_ops.col_stats(x) if stats is None else stats
    params = None if sw is None else (sw, sb, bw, bb)
    t2 = None if t is None else _f(t.reshape(B, -1).float())
    a, o = hip_ops.adagn_coeffs(stats, R, t2, params, G, eps)
    return a, o, stats, t2


def _pro_ok(R: int, K: int, Nout: int) -> bool:
    """The AdaGN-as-prologue Functions: split-bf16 training precision, shapes the LDS-DMA kernels take with a prologue."""
    prec = _lin_precision()
    if os.environ.get("GECCO_TRAIN_ADAGNPRO", "1") == "0" or prec not in ("bf16x3", "fp16") or K > 1024 or R % 32:
        return False
    lib = _lib.load()
    return bool(lib.gecco_linear_image_ok_f16(R, K, Nout, 1) if prec == "fp16" else lib.gecco_linear_image_ok(R, K, Nout, 1))


def _io16_ok(R: int, Cc: int, H: int, I: int) -> bool:
    """Under the autocast(float16) arithmetic the projections K | V and q of a layer, the unpool attention's output and their gradients
    are read only by kernels that round them to fp16 on the way in (both attentions and their backward, out_proj and its dX / dW products,
    the dX / dW products of kv_proj | q_proj): they are then fp16 TENSORS, as in the reference's autocast run (Lightning's
    precision="16-mixed": nn.Linear / SDPA outputs are halves there, example_configs/shapenet_airplane_unconditional.py:74) — the same
    operand bits, half the bytes of fifteen crossings of HBM per layer (~1.7 GB at the shipped batch).  Shapes every kernel of that
    chain takes: whole 128-row blocks, 64 inducers, head dims 16 / 32 / 48 / 64, feature_dim 128 / 256 / 384 / 512.  GECCO_TRAIN_IO16=0: fp32 tensors."""
    if os.environ.get("GECCO_TRAIN_IO16", "1") == "0" or _lin_precision() != "fp16" or os.environ.get("GECCO_TRAIN_ATTN16", "1") == "0":
        return False
    hd = Cc // H
    return (I == 64 and Cc % H == 0 and hd in (16, 32, 48, 64) and Cc in (128, 256, 384, 512) and R >= 128 and R % 128 == 0
            and _fused_attn_ok(I, hd) and _a16_ok("fp16", R, Cc, 3 * Cc) and _image_ok(R, Cc, Cc, "fp16") and _image_ok(R, 2 * Cc, Cc, "fp16"))


class AdaGNPairFn(torch.autograd.Function):
    """(KV, q, x) = (AdaGN(x) Wkv^T, AdaGN(x) Wq^T + bq, x): broadcast_norm and the two projections of its output
    (models/set_transformer.py:161-162 -> :49, :112) as ONE Function.  AdaGN(x) is never materialised: the pair GEMM applies
    a x + o on its A fragments (as the inference path does), the weight-gradient kernel applies it while staging x
    (`gecco_gemm_tn_x3_pro_f32`), and the backward continues into the AdaGN backward with the two dX products' sum.  The third
    output hands x to the skip connection; the gradient returning through it is added inside the AdaGN backward kernel."""

    @staticmethod
    def forward(ctx, x, t, sw, sb, bw, bb, G, eps, stats, W1, W2, b2, io16=False):
        x = _f(x)
        ctx.set_materialize_grads(False)
        a, o, stats, t2 = _adagn_coeffs(x, t, sw, sb, bw, bb, G, eps, stats)
        prec = ctx.prec = _lin_precision()   # "bf16x3" or "fp16" (_pro_ok)
        ctx.side2 = bool(getattr(W2, "_gecco_side_ok", False))
        B, R, K = x.shape
        N1, N2 = W1.shape[0], W2.shape[0]
        if io16:   # (`_io16_ok`) K | V and q as fp16 tensors: the same A-stationary kernel, its fp16 row-major epilogue
            ws, ready = _a16_stream("pair", W1, W2, dev=x.device)
            y16 = torch.empty(B, R, K, device=x.device, dtype=torch.float16) if _y16_ok(R, K, N1) and _y16_ok(R, K, N2) else None
            KV, q = hip_ops.linear_kvq_f16(x, (a, o), W1, None, _f(W2), b2, lo=(0, 0), head_dim=0, wsplit=ws, image_ready=ready, y16=y16)
            ctx.save_for_backward(x, stats, t2, sw, sb, a, o, W1, W2, bw, y16 if y16 is not None else x.new_empty(0))
            ctx.G, ctx.eps, ctx.has_b2, ctx.t_shape = G, eps, b2 is not None, tuple(t.shape)
            return KV, q, x
        if _a16_ok(prec, R, K, N1 + N2) and N1 % 64 == 0 and N2 % 64 == 0:
            KV, q = _new(B, R, N1, like=x), _new(B, R, N2, like=x)
            ws, ready = _a16_stream("pair", W1, W2, dev=x.device)
            _lib.check(_lib.load().gecco_linear_astat16_f32(_ptr(x), _ptr(a), _ptr(o), None if ready else _ptr(_f(W1)), None, N1, _ptr(KV),
                                                            None if ready else _ptr(_f(W2)), _ptr(b2), N2, _ptr(q), None, 0, B, R, K,
                                                            C.c_void_p(ws.data_ptr()), _stream()), "gecco_linear_astat16_f32")
            ctx.save_for_backward(x, stats, t2, sw, sb, a, o, W1, W2, bw, x.new_empty(0))
            ctx.G, ctx.eps, ctx.has_b2, ctx.t_shape = G, eps, b2 is not None, tuple(t.shape)
            return KV, q, x
        if _h8_ok(prec, R, K, N1 + N2) and N1 % 64 == 0 and N2 % 64 == 0:
            KV, q = _new(B, R, N1, like=x), _new(B, R, N2, like=x)
            ws, ready = _h8_stream("pair", W1, W2, dev=x.device)
            _lib.check(_lib.load().gecco_linear_h8_train_f32(_ptr(x), _ptr(a), _ptr(o), None if ready else _ptr(_f(W1)), None, N1, _ptr(KV),
                                                             None if ready else _ptr(_f(W2)), _ptr(b2), N2, _ptr(q), None, 0, None, B, R, K,
                                                             C.c_void_p(ws.data_ptr()), _stream()), "gecco_linear_h8_train_f32")
            ctx.save_for_backward(x, stats, t2, sw, sb, a, o, W1, W2, bw, x.new_empty(0))
            ctx.G, ctx.eps, ctx.has_b2, ctx.t_shape = G, eps, b2 is not None, tuple(t.shape)
            return KV, q, x
        img = WEIGHT_IMAGES.lookup("pair", W1, W2, prec=prec)
        if img is not None:
            KV, q = hip_ops.linear_pair(x, W1, None, W2, b2, pro=(a, o), precision=prec, w_image=img)
        else:
            KV, q = hip_ops.linear_pair(x, W1, None, _f(W2), b2, pro=(a, o), precision=prec)
        ctx.save_for_backward(x, stats, t2, sw, sb, a, o, W1, W2, bw, x.new_empty(0))
        ctx.G, ctx.eps, ctx.has_b2, ctx.t_shape = G, eps, b2 is not None, tuple(t.shape)
        return KV, q, x

    @staticmethod
    def backward(ctx, dKV, dq, dskip):
        x, stats, t2, sw, sb, a, o, W1, W2, bw, y16 = ctx.saved_tensors
        need = ctx.needs_input_grad
        B, R, Cc = x.shape
        dKV = _f(dKV) if dKV is not None else x.new_zeros(B, R, W1.shape[0])
        dq = _f(dq) if dq is not None else x.new_zeros(B, R, W2.shape[0])
        prec = ctx.prec
        gst = None
        if (dq.dtype == torch.float16 and os.environ.get("GECCO_TRAIN_DOTSTATS", "1") != "0" and os.environ.get("GECCO_TRAIN_DOTRES", "1") != "0"
                and _lib.load().gecco_linear_actbwd_ok(R, W2.shape[0], Cc, 2) and R >= 128):
            # fp16-tensor layer: the second dX product adds the first and leaves {sum dY, sum dY x} for the AdaGN backward below
            # (no col_dot_stats pass over dY and x)
            dY, gst = _linear_dx_dot(dq, W2, x, prec=prec, residual=_linear_dx(dKV, W1, prec=prec))
        else:
            dY = _linear_dx(dq, W2, residual=_linear_dx(dKV, W1, prec=prec), prec=prec)
        # X of the two weight gradients: the fp16 operand the forward stored (both operands fp16: the DMA form), else x with the AdaGN apply
        xw, prow = (y16, None) if (y16.numel() and dKV.dtype == torch.float16 and dq.dtype == torch.float16) else (x, (a, o))
        dW1 = _linear_dw(dKV, xw, pro=prow, leaf=W1, prec=prec) if need[9] else None
        dW2 = db2 = None
        if need[10] and ctx.has_b2 and need[11]:
            dW2, db2 = _linear_dw(dq, xw, want_db=True, pro=prow, leaf=W2, prec=prec, side_ok=ctx.side2)
        elif need[10]:
            dW2 = _linear_dw(dq, xw, pro=prow, leaf=W2, prec=prec, side_ok=ctx.side2)
        elif ctx.has_b2 and need[11]:
            db2 = _linear_db(dq.float() if dq.dtype != torch.float32 else dq)
        dx, dsw, dsb, dbw, dbb, dt = _adagn_backward(x, stats, t2, sw, sb, dY, dskip, ctx.G, ctx.eps, True, gst=gst, bw=bw if need[1] else None)
        return dx, (dt.reshape(ctx.t_shape) if dt is not None else None), dsw, dsb, dbw, dbb, None, None, None, dW1, dW2, db2, None


class AdaGNMlpFn(torch.autograd.Function):
    """x + MLP(AdaGN(x)) (models/set_transformer.py:165-166: mlp_norm, Linear -> act -> Linear, the skip) as ONE Function:
    AdaGN as the prologue of the first GEMM (whose epilogue keeps the pre-activation), the skip as the epilogue of the
    second (which with want_stats also leaves the next norm's partial sums); backward: act' as the epilogue of the second
    linear's dX product, the first linear's weight gradient from x through the prologue, then the AdaGN backward with the
    skip's gradient added in its kernel."""

    @staticmethod
    def forward(ctx, x, t, sw, sb, bw, bb, G, eps, stats, W0, b0, alpha, W2, b2, kind, want_stats):
        x = _f(x)
        lib = _lib.load()
        B, R, K0 = x.shape
        N0 = W0.shape[0]
        a, o, stats, t2 = _adagn_coeffs(x, t, sw, sb, bw, bb, G, eps, stats)
        prec = ctx.prec = _lin_precision()   # "bf16x3" or "fp16" (_pro_ok)
        h16 = _h16_ok(prec, R, K0, N0, W2.shape[0], True)
        y16 = None
        if h16:
            # (when the backward will hold du as halves: keep fp16(AdaGN(x)) for mlp.0's weight gradient)
            want_y = kind in (1, 2, 3) and _du16_ok(prec, R, W2.shape[0], N0, K0) and _y16_ok(R, K0, N0)
            u, h, y16 = _keep_h16(x, W0, b0, (a, o), alpha, kind, want_y16=True) if want_y else (*_keep_h16(x, W0, b0, (a, o), alpha, kind), None)
        elif kind in (1, 2, 3) and _h8_ok(prec, R, K0, N0):
            u, h = _new(B, R, N0, like=x), _new(B, R, N0, like=x)
            ws, ready = _h8_stream("n", W0, dev=x.device)
            _lib.check(lib.gecco_linear_h8_train_f32(_ptr(x), _ptr(a), _ptr(o), None if ready else _ptr(_f(W0)), _ptr(b0), N0, _ptr(h),
                                                     None, None, 0, None, _ptr(alpha) if kind in (1, 2) else None, kind, _ptr(u), B, R, K0,
                                                     C.c_void_p(ws.data_ptr()), _stream()), "gecco_linear_h8_train_f32")
        else:
            u, h = _new(B, R, N0, like=x), _new(B, R, N0, like=x)
            img = WEIGHT_IMAGES.lookup("n", W0, prec=prec)
            Wp, ws = (None, img) if img is not None else (_f(W0), hip_ops._ws((N0 + 127) // 128 * 128 * K0 * 4, x.device))
            _lib.check(lib.gecco_linear_act_keep_pro_f32(_ptr(x), _ptr(Wp), _ptr(b0), _ptr(a), _ptr(o), _ptr(alpha) if kind in (1, 2) else None,
                                                         kind, _ptr(u), _ptr(h), B, R, K0, N0, hip_ops.PRECISIONS[prec],
                                                         C.c_void_p(ws.data_ptr()), _stream()),
                       "gecco_linear_act_keep_pro_f32")
        ctx.save_for_backward(x, stats, t2, sw, sb, a, o, u, h, alpha if alpha is not None else x.new_empty(0), W0, W2, bw,
                              y16 if y16 is not None else x.new_empty(0))
        ctx.G, ctx.eps, ctx.kind, ctx.bias, ctx.t_shape = G, eps, kind, (b0 is not None, b2 is not None), tuple(t.shape)
        out = _linear_fwd_h16(h, W2, b2, x, want_stats) if h16 else _linear_fwd(h, W2, b2, x, want_stats, prec)
        if want_stats:
            ctx.mark_non_differentiable(out[1])
            ctx.set_materialize_grads(False)   # no zero-filled gradient tensor for the statistics
        return out

    @staticmethod
    def backward(ctx, dout, _dstats=None):
        if dout is None:   # (materialize off) only the statistics were used: they carry no gradient
            return (None,) * len(ctx.needs_input_grad)
        x, stats, t2, sw, sb, a, o, u, h, alpha, W0, W2, bw, y16 = ctx.saved_tensors
        need = ctx.needs_input_grad
        dout = _f(dout)
        prec = ctx.prec
        # (kind 1 - 3 with every consumer of du on an fp16 kernel: du as halves)
        du16 = (ctx.kind in (1, 2, 3) and need[0] and (need[9] or not (ctx.bias[0] and need[10]))
                and _du16_ok(prec, x.shape[1], W2.shape[0], W2.shape[1], x.shape[2]))
        du, dalpha = _act_linear_dx(dout, u, h, alpha, W2, ctx.kind, need[11], prec, du16=du16)

        def wgrads(g, act_in, has_b, iw, ib, Wl, pro=None):
            if has_b and need[ib] and need[iw]:
                return _linear_dw(g, act_in, want_db=True, pro=pro, leaf=Wl, prec=prec)
            return (_linear_dw(g, act_in, pro=pro, leaf=Wl, prec=prec) if need[iw] else None), (_linear_db(g) if has_b and need[ib] else None)
        dW2, db2 = wgrads(dout, h, ctx.bias[1], 12, 13, W2)
        if du.dtype == torch.float16 and y16.numel():   # both operands as fp16 images: the DMA form of the weight-gradient kernel
            dW0, db0 = wgrads(du, y16, ctx.bias[0], 9, 10, W0)
        else:
            dW0, db0 = wgrads(du, x, ctx.bias[0], 9, 10, W0, pro=(a, o))
        dY, gst = _linear_dx_dot(du, W0, x, prec=prec)
        dx, dsw, dsb, dbw, dbb, dt = _adagn_backward(x, stats, t2, sw, sb, dY, dout, ctx.G, ctx.eps, True, gst=gst, bw=bw if need[1] else None)
        return dx, (dt.reshape(ctx.t_shape) if dt is not None else None), dsw, dsb, dbw, dbb, None, None, None, dW0, db0, dalpha, dW2, db2, None, None


# ------------------------------------------------------------------------------------------- activation
class GaussActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, alpha, normalized):
        u = _f(u)
        ctx.save_for_backward(u, alpha)
        ctx.normalized = normalized
        return hip_ops.gaussian_act(u, alpha, normalized)

    @staticmethod
    def backward(ctx, dy):
        u, alpha = ctx.saved_tensors
        dy = _f(dy)
        lib = _lib.load()
        n = u.numel()
        nb = lib.gecco_gauss_act_bwd_blocks(n)
        du, part = torch.empty_like(u), _new(nb, like=u)
        _lib.check(lib.gecco_gauss_act_bwd_f32(_ptr(u), _ptr(dy), _ptr(alpha), _ptr(du), _ptr(part), n, int(ctx.normalized),
                                               _stream()), "gauss_act_bwd")
        dalpha = _reduce(part, 1, nb, 1).reshape(alpha.shape)
        return du, dalpha, None


def _du16_ok(prec: str, R: int, Nout: int, K: int, K0: int) -> bool:
    """Under the autocast(float16) arithmetic the gradient du = (dy W2) act'(u) of an MLP's hidden pre-activation is read again only by
    the matrix pipe — the first linear's weight gradient and its dX product — as an fp16 operand either way (the reference's autocast
    backward holds it as an fp16 tensor: autograd of models/mlp.py:5-39 under precision="16-mixed"): the dX product's epilogue stores it
    as halves, half the bytes of its three crossings of HBM, and the dX product that reads it runs on fp16 tiles (2 x fewer bytes into
    the CU per FLOP).  Same operand bits as rounding the fp32 tensor at the consumers; the bias gradient (column sums of du) is then formed
    from the halves.  (R, Nout -> K): the product that forms du; K0: the first linear's input width.  GECCO_TRAIN_DU16=0: fp32 du."""
    if (prec != "fp16" or os.environ.get("GECCO_TRAIN_DU16", "1") == "0" or os.environ.get("GECCO_TRAIN_ACTBWD", "1") == "0"
            or os.environ.get("GECCO_TRAIN_DOTSTATS", "1") == "0" or not _a16_ok(prec, R, Nout, K)):
        return False
    lib = _lib.load()
    return bool(K % 32 == 0 and R >= 128 and R % 32 == 0 and K0 % 4 == 0 and lib.gecco_linear_actbwd_ok(R, Nout, K, 2)
                and lib.gecco_linear_actbwd_ok(R, K, K0, 2))


def _act_linear_dx(dy: Tensor, u: Tensor, h: Tensor, alpha: Tensor, W: Tensor, kind: int, want_alpha: bool, prec: str | None = None,
                   du16: bool = False):
    """du = (dy W) * act'(u) and, for GaussianActivation, d alpha: the activation's backward as the epilogue of the dX product
    (`gecco_linear_actbwd_f32`) where the LDS-DMA kernels take the shape, else the product followed by the activation's
    backward kernel.  du16 (the caller checked `_du16_ok`): du leaves as an fp16 tensor."""
    lib = _lib.load()
    B, R, Nout = dy.shape
    K = W.shape[1]
    prec = _resolve(prec, R, Nout, K)
    dalpha = None
    fused = (os.environ.get("GECCO_TRAIN_ACTBWD", "1") != "0" and prec in ("fp32", "bf16x3", "fp16")
             and lib.gecco_linear_actbwd_ok(R, Nout, K, hip_ops.PRECISIONS[prec]))
    if fused and kind in (1, 2, 3) and _a16_ok(prec, R, Nout, K):
        du = torch.empty(u.shape, device=u.device, dtype=torch.float16) if du16 else torch.empty_like(u)
        parts = _new(B * R // 128, like=u) if kind in (1, 2) else None
        ws, ready = _a16_stream("t", W, dev=u.device)
        if du16:
            _lib.check(lib.gecco_linear_astat16_actbwd_h16(_ptr(dy), None if ready else _ptr(_f(W)), _ptr(u), _ptr(alpha) if kind in (1, 2) else None,
                                                           kind, C.c_void_p(du.data_ptr()), _ptr(parts), B, R, Nout, K, C.c_void_p(ws.data_ptr()),
                                                           _stream()), "gecco_linear_astat16_actbwd_h16")
        else:
            _lib.check(lib.gecco_linear_astat16_actbwd(_ptr(dy), None if ready else _ptr(_f(W)), _ptr(u), _ptr(alpha) if kind in (1, 2) else None, kind,
                                                       _ptr(du), _ptr(parts), B, R, Nout, K, C.c_void_p(ws.data_ptr()), _stream()),
                       "gecco_linear_astat16_actbwd")
        if kind in (1, 2) and want_alpha:
            dalpha = _reduce(parts, 1, B * R // 128, 1).reshape(alpha.shape)
        return du, dalpha
    assert not du16, "_du16_ok admits only the A-stationary activation-backward kernel"
    if fused:
        du = torch.empty_like(u)
        nt = lib.gecco_linear_actbwd_tiles(B, R, K)
        parts = torch.zeros(nt, device=u.device, dtype=torch.float32) if kind in (1, 2) else None
        img = WEIGHT_IMAGES.lookup("t", W, prec=prec) if prec in ("bf16x3", "fp16") else None
        if img is not None:
            Wt, ws = None, img
        else:
            Wt = W.t().contiguous()
            ws = hip_ops._ws((K + 127) // 128 * 128 * Nout * 4, u.device) if prec != "fp32" else None
        _lib.check(lib.gecco_linear_actbwd_f32(_ptr(dy), _ptr(Wt), _ptr(u), _ptr(alpha) if kind in (1, 2) else None, kind, None,
                                               _ptr(du), _ptr(parts), B, R, Nout, K, hip_ops.PRECISIONS[prec],
                                               C.c_void_p(ws.data_ptr()) if ws is not None else None, _stream()),
                   "gecco_linear_actbwd_f32")
        if kind in (1, 2) and want_alpha:
            dalpha = _reduce(parts, 1, nt, 1).reshape(alpha.shape)
        return du, dalpha
    dh = _linear_dx(dy, W, prec=prec)
    if kind in (1, 2):
        nb = lib.gecco_gauss_act_bwd_blocks(u.numel())
        du, part = torch.empty_like(u), _new(nb, like=u)
        _lib.check(lib.gecco_gauss_act_bwd_f32(_ptr(u), _ptr(dh), _ptr(alpha), _ptr(du), _ptr(part), u.numel(), int(kind == 1),
                                               _stream()), "gauss_act_bwd")
        dalpha = _reduce(part, 1, nb, 1).reshape(alpha.shape)
    elif kind == 3:
        du = hip_ops.relu_bwd(h if h.dtype == torch.float32 else u, dh)   # (u > 0) == (relu(u) > 0)
    else:
        du = torch.empty_like(u)
        _lib.check(lib.gecco_gelu_bwd_f32(_ptr(u), _ptr(dh), _ptr(du), u.numel(), _stream()), "gecco_gelu_bwd_f32")
    return du, dalpha


def _act_forward(u: Tensor, alpha: Tensor | None, kind: int) -> Tensor:
    if kind in (1, 2):
        return hip_ops.gaussian_act(u, alpha, kind == 1)
    if kind == 3:
        return hip_ops.relu(u)
    h = torch.empty_like(u)
    _lib.check(_lib.load().gecco_gelu_f32(_ptr(u), _ptr(h), u.numel(), _stream()), "gecco_gelu_f32")
    return h


def _linear_fwd(x: Tensor, W: Tensor, b, res, want_stats: bool, prec: str | None = None):
    """hip_ops.linear in the training precision (prec: the Function's `_lin_precision()`), with the step's ready weight image when
    there is one."""
    prec = _resolve(prec, x.shape[1], W.shape[1], W.shape[0])
    img = WEIGHT_IMAGES.lookup("n", W, prec=prec) if _image_ok(x.shape[1], W.shape[1], W.shape[0], prec) else None
    kw = dict(precision=prec, w_image=img, w_shape=tuple(W.shape)) if img is not None else dict(precision=prec)
    return hip_ops.linear(x, None if img is not None else W, b, residual=res, want_stats=want_stats, **kw)


def _a16_ok(prec: str, R: int, K: int, Nout: int) -> bool:
    """The A-stationary fp16 kernels (gecco_linear_astat16_*) take this product: the autocast(float16) arithmetic, whole 128-row blocks,
    an operand of <= 512 columns held in registers, 64-column output tiles.  GECCO_TRAIN_A16=0: the LDS-DMA GEMM instead."""
    return (prec == "fp16" and os.environ.get("GECCO_TRAIN_A16", "1") != "0"
            and bool(_lib.load().gecco_linear_astat16_ok(R, K, Nout)))


def _a16_stream(kind: str, *ws: Tensor, dev) -> tuple[Tensor, bool]:
    """(wsplit, ready): the step's batched stream of these weights if it is there, else scratch the entry point fills itself."""
    img = WEIGHT_IMAGES.lookup(kind, *ws, prec="a16")
    if img is not None:
        return img, True
    lib = _lib.load()
    tr = kind == "t"
    nbytes = sum(lib.gecco_astat16_image_bytes(w.shape[1] if tr else w.shape[0], w.shape[0] if tr else w.shape[1]) for w in ws)
    return hip_ops._ws(nbytes, dev), False


def _h8_ok(prec: str, R: int, K: int, Nout: int) -> bool:
    """The split-bf16 training FORWARD's AdaGN-prologue products on the h8 A-stationary kernel (gecco_linear_h8_train_f32): the same
    accuracy class as split-bf16 (fp16 main product + two fp8 cross terms), the operand rows in registers and the weight stream read
    once per 128 rows.  Forward only: activations and weights fit the fp16 / fp8 operand ranges, unscaled gradients do not, so the
    backward products keep split-bf16.  Only where the model's arithmetic is one of the modes that compute these very products that way
    in inference ("mixed", "w2"; "fp16"): an explicit "bf16x3" keeps true split-bf16 (fp32 exponent range, no operand clamps) in the
    forward too.  GECCO_TRAIN_H8FWD=0: the LDS-DMA split-bf16 GEMM everywhere."""
    return (prec == "bf16x3" and hip_ops.default_precision() in ("mixed", "w2", "fp16") and os.environ.get("GECCO_TRAIN_H8FWD", "1") != "0"
            and bool(_lib.load().gecco_linear_h8_train_ok(R, K, Nout)))


def _h8_stream(kind: str, *ws: Tensor, dev) -> tuple[Tensor, bool]:
    img = WEIGHT_IMAGES.lookup(kind, *ws, prec="h8")
    if img is not None:
        return img, True
    lib = _lib.load()
    return hip_ops._ws(sum(lib.gecco_h8_image_bytes(w.shape[0], w.shape[1]) for w in ws), dev), False


def _h16_ok(prec: str, R: int, K0: int, N0: int, Nout2: int, pro: bool) -> bool:
    """Under the autocast(float16) arithmetic the hidden layer h = act(u) of an MLP is read again only by the matrix pipe — the
    second linear's forward and its weight gradient — as an fp16 operand either way (the reference's autocast stores it as fp16
    too): the first GEMM's epilogue stores it as fp16 (`gecco_linear_act_keep_h16`), half the bytes of its three crossings of HBM.
    u (the pre-activation the backward differentiates) stays fp32."""
    if prec != "fp16" or os.environ.get("GECCO_TRAIN_H16", "1") == "0" or R < 128 or R % 32 or N0 % 8 or Nout2 % 4:
        return False
    lib = _lib.load()
    return bool(lib.gecco_linear_image_ok_f16(R, K0, N0, int(pro)) and lib.gecco_linear_image_ok_f16(R, N0, Nout2, 0))


def _y16_ok(R: int, K: int, Nout: int) -> bool:
    """The weight gradient dY^T X of a linear whose dY is an fp16 tensor (`_io16_ok`, `_du16_ok`) on fp16 images of BOTH operands: the
    forward kernel stores the operand it forms, y16 = fp16(AdaGN(x)), and the weight-gradient kernel takes its slabs global -> LDS by DMA
    (gemm_tn_f16_dma_kernel: whole 128 x 128 tiles).  Default off: X = x with the AdaGN apply while it is staged."""
    # OPT-IN (measured and lost, profiles/r06_negative_results.txt item 10): the kernel alone 168 -> 143 us, the step 16.77 -> 17.26 ms
    return os.environ.get("GECCO_TRAIN_Y16", "0") == "1" and R % 32 == 0 and K % 128 == 0 and Nout % 128 == 0


def _keep_h16(x: Tensor, W0: Tensor, b0, pro, alpha, kind: int, want_y16: bool = False):
    """(u fp32, h fp16) = (x' W0^T + b0, act(u)) from one launch; pro = (a, o): x' = a x + o.  want_y16 (A-stationary kernel only): also
    y16 = fp16(x') -> (u, h, y16 | None)."""
    lib = _lib.load()
    B, R, K0 = x.shape
    N0 = W0.shape[0]
    u, h = _new(B, R, N0, like=x), torch.empty(B, R, N0, device=x.device, dtype=torch.float16)
    if kind in (1, 2, 3) and _a16_ok("fp16", R, K0, N0):
        ws, ready = _a16_stream("n", W0, dev=x.device)
        y16 = torch.empty(B, R, K0, device=x.device, dtype=torch.float16) if want_y16 else None
        _lib.check(lib.gecco_linear_astat16_keep_y16(_ptr(x), _ptr(pro[0]) if pro else None, _ptr(pro[1]) if pro else None,
                                                     None if ready else _ptr(_f(W0)), _ptr(b0), _ptr(alpha) if kind in (1, 2) else None, kind,
                                                     _ptr(u), C.c_void_p(h.data_ptr()), C.c_void_p(y16.data_ptr()) if want_y16 else None,
                                                     B, R, K0, N0, C.c_void_p(ws.data_ptr()), _stream()),
                   "gecco_linear_astat16_keep_y16")
        return (u, h, y16) if want_y16 else (u, h)
    if want_y16:
        return (*_keep_h16(x, W0, b0, pro, alpha, kind), None)
    img = WEIGHT_IMAGES.lookup("n", W0, prec="fp16")
    Wp, ws = (None, img) if img is not None else (_f(W0), hip_ops._ws((N0 + 127) // 128 * 128 * K0 * 2, x.device))
    _lib.check(lib.gecco_linear_act_keep_h16(_ptr(x), _ptr(Wp), _ptr(b0), _ptr(pro[0]) if pro else None, _ptr(pro[1]) if pro else None,
                                             _ptr(alpha) if kind in (1, 2) else None, kind, _ptr(u), C.c_void_p(h.data_ptr()), B, R, K0, N0,
                                             C.c_void_p(ws.data_ptr()), _stream()), "gecco_linear_act_keep_h16")
    return u, h


def _linear_fwd_h16(h: Tensor, W: Tensor, b, res, want_stats: bool):
    """The second linear of such an MLP: fp16 A tensor, fp32 output (+ residual, + statistics)."""
    img = WEIGHT_IMAGES.lookup("n", W, prec="fp16")
    return hip_ops.linear_f16io(h, None if img is not None else W, b, residual=res, want_stats=want_stats, w_image=img,
                                w_shape=tuple(W.shape))


class LinearActLinearFn(torch.autograd.Function):
    """y = act(x @ W0^T + b0) @ W2^T + b2 (+ residual): an MLP of the reference (models/mlp.py: Linear -> act -> Linear; a
    CNBlock's pointwise pair) as ONE Function.  Forward: the first GEMM's epilogue leaves both u = x W0^T + b0 and act(u)
    (`gecco_linear_act_keep_f32`: no activation pass); backward: act' is the epilogue of the second linear's dX product
    (`_act_linear_dx`).  kind: 1 / 2 GaussianActivation normalized / raw, 3 ReLU, 4 GELU."""

    @staticmethod
    def forward(ctx, x, W0, b0, alpha, W2, b2, residual, kind, want_stats=False):
        x = _f(x)
        lib = _lib.load()
        B, R, K0 = x.shape
        N0 = W0.shape[0]
        prec = ctx.prec = _lin_precision()
        keep = (os.environ.get("GECCO_TRAIN_ACTKEEP", "1") != "0" and prec in ("fp32", "bf16x3", "fp16")
                and lib.gecco_linear_actbwd_ok(R, K0, N0, hip_ops.PRECISIONS[prec]))
        h16 = keep and _h16_ok(prec, R, K0, N0, W2.shape[0], False)
        if h16:
            u, h = _keep_h16(x, W0, b0, None, alpha, kind)
        elif keep:
            u, h = _new(B, R, N0, like=x), _new(B, R, N0, like=x)
            img = WEIGHT_IMAGES.lookup("n", W0, prec=prec) if prec in ("bf16x3", "fp16") and _image_ok(R, K0, N0, prec) else None
            if img is not None:
                Wp, ws = None, img
            else:
                Wp = _f(W0)
                ws = hip_ops._ws((N0 + 127) // 128 * 128 * K0 * 4, x.device) if prec != "fp32" else None
            _lib.check(lib.gecco_linear_act_keep_f32(_ptr(x), _ptr(Wp), _ptr(b0), _ptr(alpha) if kind in (1, 2) else None, kind,
                                                     _ptr(u), _ptr(h), B, R, K0, N0, hip_ops.PRECISIONS[prec],
                                                     C.c_void_p(ws.data_ptr()) if ws is not None else None, _stream()),
                       "gecco_linear_act_keep_f32")
        else:
            u = _linear_fwd(x, W0, b0, None, False, prec)
            h = _act_forward(u, alpha, kind)
        ctx.save_for_backward(x, u, h, alpha if alpha is not None else x.new_empty(0), W0, W2)
        ctx.kind, ctx.bias = kind, (b0 is not None, b2 is not None)
        res = None if residual is None else _f(residual)
        out = _linear_fwd_h16(h, W2, b2, res, want_stats) if h16 else _linear_fwd(h, W2, b2, res, want_stats, prec)
        if want_stats:
            ctx.mark_non_differentiable(out[1])
            ctx.set_materialize_grads(False)   # no zero-filled gradient tensor for the statistics
        return out

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        if dy is None:   # (materialize off) only the statistics were used: they carry no gradient
            return (None,) * len(ctx.needs_input_grad)
        x, u, h, alpha, W0, W2 = ctx.saved_tensors
        need = ctx.needs_input_grad
        dy = _f(dy)
        prec = ctx.prec
        du, dalpha = _act_linear_dx(dy, u, h, alpha, W2, ctx.kind, need[3], prec)

        def wgrads(g, a, has_b, iw, ib, Wl):
            if has_b and need[ib] and need[iw]:
                return _linear_dw(g, a, want_db=True, leaf=Wl, prec=prec)
            return (_linear_dw(g, a, leaf=Wl, prec=prec) if need[iw] else None), (_linear_db(g) if has_b and need[ib] else None)
        dW2, db2 = wgrads(dy, h, ctx.bias[1], 4, 5, W2)
        dW0, db0 = wgrads(du, x, ctx.bias[0], 1, 2, W0)
        dx = _linear_dx(du, W0, prec=prec) if need[0] else None
        return dx, dW0, db0, dalpha, dW2, db2, (dy if need[6] else None), None, None


class ActLinearFn(torch.autograd.Function):
    """y = act(u) @ W^T + b (+ residual): the activation and the linear that follows it (models/mlp.py: Linear -> act ->
    Linear; a CNBlock's Linear -> GELU -> Linear) as ONE Function, so that the backward can run the activation's derivative
    as the EPILOGUE of the dX product (`gecco_linear_actbwd_f32`): dh = dy W never exists, du leaves the GEMM, and for
    GaussianActivation the alpha gradient's partial sums come out of the same epilogue.  kind: 1 / 2 GaussianActivation
    normalized / raw, 3 ReLU, 4 GELU.  Shapes the LDS-DMA kernels do not take fall back to the two-kernel backward."""

    @staticmethod
    def forward(ctx, u, alpha, W, b, residual, kind, want_stats=False):
        u = _f(u)
        h = _act_forward(u, alpha, kind)
        ctx.save_for_backward(u, h, alpha if alpha is not None else u.new_empty(0), W)
        ctx.kind, ctx.has_bias = kind, b is not None
        prec = ctx.prec = _lin_precision()
        res = None if residual is None else _f(residual)
        out = _linear_fwd(h, W, b, res, want_stats, prec)
        if want_stats:
            ctx.mark_non_differentiable(out[1])
            ctx.set_materialize_grads(False)   # no zero-filled gradient tensor for the statistics
        return out

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        if dy is None:   # (materialize off) only the statistics were used: they carry no gradient
            return (None,) * len(ctx.needs_input_grad)
        u, h, alpha, W = ctx.saved_tensors
        dy = _f(dy)
        prec = ctx.prec
        du, dalpha = _act_linear_dx(dy, u, h, alpha, W, ctx.kind, ctx.needs_input_grad[1], prec)
        dW = db = None
        if ctx.has_bias and ctx.needs_input_grad[3] and ctx.needs_input_grad[2]:
            dW, db = _linear_dw(dy, h, want_db=True, leaf=W, prec=prec)
        elif ctx.needs_input_grad[2]:
            dW = _linear_dw(dy, h, leaf=W, prec=prec)
        elif ctx.has_bias and ctx.needs_input_grad[3]:
            db = _linear_db(dy)
        dres = dy if ctx.needs_input_grad[4] else None
        return du, dalpha, dW, db, dres, None, None


class ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u):
        y = hip_ops.relu(_f(u))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return hip_ops.relu_bwd(y, _f(dy))


# ------------------------------------------------------------------------------------------- attention
def _softmax(S: Tensor, scale: float) -> Tensor:
    lib = _lib.load()
    P = torch.empty_like(S)
    _lib.check(lib.gecco_softmax_fwd_f32(_ptr(S), _ptr(P), S.numel() // S.shape[-1], S.shape[-1], scale, _stream()), "softmax_fwd")
    return P


def _softmax_bwd(P: Tensor, dP: Tensor, scale: float) -> Tensor:
    lib = _lib.load()
    dS = torch.empty_like(P)
    _lib.check(lib.gecco_softmax_bwd_f32(_ptr(P), _ptr(dP), _ptr(dS), P.numel() // P.shape[-1], P.shape[-1], scale, _stream()),
               "softmax_bwd")
    return dS


def _attn_pr() -> int:
    """Arithmetic code of the fused attention kernels (forward and the backward that recomputes P with the same operands): the
    training precision, or — under autocast(float16), where torch runs scaled_dot_product_attention / nn.MultiheadAttention and their
    backward with fp16 operands — 2: one fp16 plane per operand, one MFMA per product (GECCO_TRAIN_ATTN16=0: split-bf16 there too)."""
    if _lin_precision() == "fp16" and os.environ.get("GECCO_TRAIN_ATTN16", "1") != "0":
        return 2
    return min(hip_ops.PRECISIONS[_train_precision()], 1)


def _fused_attn_ok(I: int, hd: int) -> bool:
    """Shapes the fused attention kernels take (every shipped config: 64 inducers, head dim d / 8); others run the
    strided-batched GEMM form below."""
    return I == 64 and hd % 8 == 0 and 8 <= hd <= 64 and os.environ.get("GECCO_TRAIN_ATTN", "fused") != "gemm"


class PoolAttnFn(torch.autograd.Function):
    """AttentionPool core: KV (B, N, 2C), inducers (1, H, I, hd) -> merged heads (B, I, C)."""

    @staticmethod
    def forward(ctx, KV, ind, H):
        KV, ind = _f(KV), _f(ind)
        B, N, C2 = KV.shape
        Cc, I, hd = C2 // 2, ind.shape[2], ind.shape[3]
        ctx.H = H
        ctx.fused = _fused_attn_ok(I, hd)
        if KV.dtype == torch.float16:   # an `_io16_ok` layer: K | V as an fp16 tensor (row-major), the fp16 kernels
            assert ctx.fused
            lib = _lib.load()
            nb = lib.gecco_pool_attn_workspace_bytes(B, N, Cc, H, I)
            ws = torch.empty(nb, dtype=torch.uint8, device=KV.device)
            O = _new(B, I, Cc, like=ind)
            ctx.pr = 3
            _lib.check(lib.gecco_pool_attn_f16in(hip_ops._ptr16(KV), _ptr(ind), _ptr(O), B, N, Cc, H, I, 0, C.c_void_p(ws.data_ptr()), nb,
                                                 _stream()), "gecco_pool_attn_f16in")
            lse = _new(B, H, I, like=ind)
            _lib.check(lib.gecco_pool_attn_lse_f32(C.c_void_p(ws.data_ptr()), nb, _ptr(lse), B, N, Cc, H, I, _stream()),
                       "gecco_pool_attn_lse_f32")
            ctx.save_for_backward(KV, ind, O, lse)
            return O
        if ctx.fused:
            # flash-style forward (no (B, H, I, N) tensor) + the log-sum-exp the fused backward recomputes P from
            lib = _lib.load()
            nb = lib.gecco_pool_attn_workspace_bytes(B, N, Cc, H, I)
            ws = torch.empty(nb, dtype=torch.uint8, device=KV.device)
            O = _new(B, I, Cc, like=KV)
            pr = ctx.pr = _attn_pr()
            _lib.check(lib.gecco_pool_attn_ex_f32(_ptr(KV), _ptr(ind), _ptr(O), B, N, Cc, H, I, pr, C.c_void_p(ws.data_ptr()), nb,
                                                  _stream()), "gecco_pool_attn_ex_f32")
            lse = _new(B, H, I, like=KV)
            _lib.check(lib.gecco_pool_attn_lse_f32(C.c_void_p(ws.data_ptr()), nb, _ptr(lse), B, N, Cc, H, I, _stream()),
                       "gecco_pool_attn_lse_f32")
            ctx.save_for_backward(KV, ind, O, lse)
            return O
        sc = 1.0 / math.sqrt(hd)
        S = _new(B, H, I, N, like=KV)
        _gemm(ind, KV, S, Z=B * H, zdiv=H, M=I, N=N, K=hd, lda=hd, ldb=C2, ldc=N, sA=(0, I * hd), sB=(N * C2, hd),
              sC=(H * I * N, I * N))
        P = _softmax(S, sc)
        O = _new(B, I, Cc, like=KV)
        _gemm(P, KV, O, Z=B * H, zdiv=H, M=I, N=hd, K=N, lda=N, ldb=C2, ldc=Cc, sA=(H * I * N, I * N), sB=(N * C2, hd),
              sC=(I * Cc, hd), b_km=True, b_off=Cc)
        ctx.save_for_backward(KV, ind, P)
        return O

    @staticmethod
    def backward(ctx, dO):
        dO = _f(dO)
        H = ctx.H
        if ctx.fused:
            KV, ind, O, lse = ctx.saved_tensors
            B, N, C2 = KV.shape
            Cc, I, hd = C2 // 2, ind.shape[2], ind.shape[3]
            lib = _lib.load()
            P = B * lib.gecco_pool_attn_bwd_partials(B, N, H)
            dKV, dQp = torch.empty_like(KV), _new(P, H, I, hd, like=ind)
            pk = hip_ops._ptr16 if KV.dtype == torch.float16 else _ptr      # (precision 3: KV and dKV are fp16 tensors)
            _lib.check(lib.gecco_pool_attn_bwd_ex_f32(pk(KV), _ptr(ind), _ptr(O), _ptr(lse), _ptr(dO), pk(dKV), _ptr(dQp), B, N, Cc,
                                                      H, I, ctx.pr, _stream()), "gecco_pool_attn_bwd_ex_f32")
            return dKV, _reduce(dQp, H * I * hd, P, H * I * hd).reshape(ind.shape), None
        KV, ind, P = ctx.saved_tensors
        B, N, C2 = KV.shape
        Cc, I, hd = C2 // 2, ind.shape[2], ind.shape[3]
        sc = 1.0 / math.sqrt(hd)
        zP, zKV = (H * I * N, I * N), (N * C2, hd)
        dKV = torch.empty_like(KV)
        # dV[n, d] = sum_i P[i, n] dO[i, d]
        _gemm(P, dO, dKV, Z=B * H, zdiv=H, M=N, N=hd, K=I, lda=N, ldb=Cc, ldc=C2, sA=zP, sB=(I * Cc, hd), sC=zKV,
              a_km=True, b_km=True, c_off=Cc)
        # dP[i, n] = sum_d dO[i, d] V[n, d]
        dP = torch.empty_like(P)
        _gemm(dO, KV, dP, Z=B * H, zdiv=H, M=I, N=N, K=hd, lda=Cc, ldb=C2, ldc=N, sA=(I * Cc, hd), sB=zKV, sC=zP, b_off=Cc)
        dS = _softmax_bwd(P, dP, sc)
        # dK[n, d] = sum_i dS[i, n] Q[i, d]
        _gemm(dS, ind, dKV, Z=B * H, zdiv=H, M=N, N=hd, K=I, lda=N, ldb=hd, ldc=C2, sA=zP, sB=(0, I * hd), sC=zKV,
              a_km=True, b_km=True)
        # dQ[i, d] = sum_b sum_n dS[i, n] K[n, d]
        dQp = _new(B, H, I, hd, like=KV)
        _gemm(dS, KV, dQp, Z=B * H, zdiv=H, M=I, N=hd, K=N, lda=N, ldb=C2, ldc=hd, sA=zP, sB=zKV, sC=(H * I * hd, I * hd),
              b_km=True)
        dind = _reduce(dQp, H * I * hd, B, H * I * hd).reshape(ind.shape)
        return dKV, dind, None


class UnpoolAttnFn(torch.autograd.Function):
    """MultiheadAttention core: q (B, N, C), kvh (B, I, 2C) -> (B, N, C)."""

    @staticmethod
    def forward(ctx, q, kvh, H):
        q, kvh = _f(q), _f(kvh)
        B, N, Cc = q.shape
        I, hd = kvh.shape[1], Cc // H
        ctx.H = H
        ctx.fused = _fused_attn_ok(I, hd)
        if q.dtype == torch.float16:   # an `_io16_ok` layer: q in, the attention output out as fp16 tensors (row-major)
            assert ctx.fused
            ctx.save_for_backward(q, kvh)
            ctx.pr = 3
            return hip_ops.unpool_attn_f16io(q, kvh, H)
        if ctx.fused:
            ctx.save_for_backward(q, kvh)
            ctx.pr = _attn_pr()
            return hip_ops.unpool_attn(q, kvh, H, precision="fp16" if ctx.pr == 2 else _train_precision())
        sc = 1.0 / math.sqrt(hd)
        S = _new(B, H, N, I, like=q)
        zq, zk, zS = (N * Cc, hd), (I * 2 * Cc, hd), (H * N * I, N * I)
        _gemm(q, kvh, S, Z=B * H, zdiv=H, M=N, N=I, K=hd, lda=Cc, ldb=2 * Cc, ldc=I, sA=zq, sB=zk, sC=zS)
        P = _softmax(S, sc)
        O = torch.empty_like(q)
        _gemm(P, kvh, O, Z=B * H, zdiv=H, M=N, N=hd, K=I, lda=I, ldb=2 * Cc, ldc=Cc, sA=zS, sB=zk, sC=zq, b_km=True, b_off=Cc)
        ctx.save_for_backward(q, kvh, P)
        return O

    @staticmethod
    def backward(ctx, dO):
        dO = _f(dO)
        H = ctx.H
        if ctx.fused:
            q, kvh = ctx.saved_tensors
            B, N, Cc = q.shape
            I = kvh.shape[1]
            lib = _lib.load()
            P = lib.gecco_unpool_attn_bwd_partials(B, N, H)
            dq, parts = torch.empty_like(q), _new(P, B, I, 2 * Cc, like=kvh)
            if q.dtype == torch.float16:   # precision 3: q, dO and dq are fp16 tensors
                dO = dO if dO.dtype == torch.float16 else dO.half()
                pk = hip_ops._ptr16
            else:
                pk = _ptr
            _lib.check(lib.gecco_unpool_attn_bwd_ex_f32(pk(q), _ptr(kvh), pk(dO), pk(dq), _ptr(parts), B, N, Cc, H, I, ctx.pr,
                                                        _stream()), "gecco_unpool_attn_bwd_ex_f32")
            n = B * I * 2 * Cc
            return dq, (parts[0] if P == 1 else _reduce(parts, n, P, n).reshape(B, I, 2 * Cc)), None
        q, kvh, P = ctx.saved_tensors
        B, N, Cc = q.shape
        I, hd = kvh.shape[1], Cc // H
        sc = 1.0 / math.sqrt(hd)
        zq, zk, zS = (N * Cc, hd), (I * 2 * Cc, hd), (H * N * I, N * I)
        dkvh = torch.empty_like(kvh)
        # dv[i, d] = sum_n P[n, i] dO[n, d]
        _gemm(P, dO, dkvh, Z=B * H, zdiv=H, M=I, N=hd, K=N, lda=I, ldb=Cc, ldc=2 * Cc, sA=zS, sB=zq, sC=zk, a_km=True,
              b_km=True, c_off=Cc)
        # dP[n, i] = sum_d dO[n, d] v[i, d]
        dP = torch.empty_like(P)
        _gemm(dO, kvh, dP, Z=B * H, zdiv=H, M=N, N=I, K=hd, lda=Cc, ldb=2 * Cc, ldc=I, sA=zq, sB=zk, sC=zS, b_off=Cc)
        dS = _softmax_bwd(P, dP, sc)
        # dq[n, d] = sum_i dS[n, i] k[i, d]
        dq = torch.empty_like(q)
        _gemm(dS, kvh, dq, Z=B * H, zdiv=H, M=N, N=hd, K=I, lda=I, ldb=2 * Cc, ldc=Cc, sA=zS, sB=zk, sC=zq, b_km=True)
        # dk[i, d] = sum_n dS[n, i] q[n, d]
        _gemm(dS, q, dkvh, Z=B * H, zdiv=H, M=I, N=hd, K=N, lda=I, ldb=Cc, ldc=2 * Cc, sA=zS, sB=zq, sC=zk, a_km=True, b_km=True)
        return dq, dkvh, None


# ------------------------------------------------------------------------------------------- lift / lower
class LiftFn(torch.autograd.Function):
    """Linear(3 -> C) on (B, N, 3) (linear_lift.py:44-46 `lift`, models/ray.py `xyz_embed`).  The gradient with respect to the geometry
    (dx = dy W: a Linear(C -> 3) with weight W^T, the lowering kernel) is formed only when a caller asks for it — guidance, score
    Jacobians; the training step never does."""

    @staticmethod
    def forward(ctx, x, W, b):
        x = _f(x)
        ctx.save_for_backward(x, W)
        ctx.C = W.shape[0]
        return hip_ops.lift(x, None, W, b)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = _f(dy)
        lib = _lib.load()
        B, N, _ = x.shape
        Cc = ctx.C
        dx = None
        if ctx.needs_input_grad[0]:
            one, zero = torch.ones(B, Cc, device=dy.device), torch.zeros(B, Cc, device=dy.device)
            dx = hip_ops.lower_edm(dy, None, None, W.t().contiguous(), torch.zeros(3, device=dy.device), gn=(one, zero))
        if not (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            return dx, None, None
        T = lib.gecco_stats_row_tiles(N)
        part = _new(B, T, 4, Cc, like=x)
        _lib.check(lib.gecco_lift_bwd_f32(_ptr(dy), _ptr(x), _ptr(part), B, N, Cc, _stream()), "lift_bwd")
        red = _reduce(part, 4 * Cc, B * T, 4 * Cc).reshape(4, Cc)
        return dx, red[:3].t().contiguous(), red[3].contiguous()


class LowerFn(torch.autograd.Function):
    """F = Linear(C -> 3)(LayerNorm_C(feat)) on (B, N, C)."""

    @staticmethod
    def forward(ctx, feat, W, b, eps):
        feat = _f(feat)
        ctx.save_for_backward(feat, W)
        ctx.eps = eps
        return hip_ops.lower_edm(feat, None, None, W, b, eps=eps)

    @staticmethod
    def backward(ctx, dF):
        feat, W = ctx.saved_tensors
        dF = _f(dF)
        lib = _lib.load()
        B, N, Cc = feat.shape
        rows = B * N
        nb = lib.gecco_lower_bwd_blocks(rows)
        dfeat, part = torch.empty_like(feat), _new(nb, 3 * Cc + 4, like=feat)
        _lib.check(lib.gecco_lower_bwd_f32(_ptr(feat), _ptr(dF), _ptr(W), _ptr(dfeat), _ptr(part), rows, Cc, ctx.eps,
                                           _stream()), "lower_bwd")
        red = _reduce(part, 3 * Cc + 4, nb, 3 * Cc + 4)
        return dfeat, red[: 3 * Cc].reshape(3, Cc), red[3 * Cc: 3 * Cc + 3].contiguous(), None


class Linear3Fn(torch.autograd.Function):
    """F = Linear(C -> 3)(y) on (B, N, C) (RayNetwork.output_proj[1], reference models/ray.py:56-59)."""

    @staticmethod
    def forward(ctx, y, W, b):
        y = _f(y)
        ctx.save_for_backward(y, W)
        B, _, Cc = y.shape
        one, zero = torch.ones(B, Cc, device=y.device), torch.zeros(B, Cc, device=y.device)
        return hip_ops.lower_edm(y, None, None, W, b, gn=(one, zero))

    @staticmethod
    def backward(ctx, dF):
        y, W = ctx.saved_tensors
        dF = _f(dF)
        lib = _lib.load()
        B, N, Cc = y.shape
        # dy = dF W  (a Linear(3 -> C) with weight W^T: the lift kernel);  dW[o, c] = sum_n dF[n, o] y[n, c] is the
        # lift's weight-gradient kernel with the roles of its two inputs exchanged
        dy = hip_ops.lift(dF, None, W.t().contiguous(), None)
        T = lib.gecco_stats_row_tiles(N)
        part = _new(B, T, 4, Cc, like=y)
        _lib.check(lib.gecco_lift_bwd_f32(_ptr(y), _ptr(dF), _ptr(part), B, N, Cc, _stream()), "lift_bwd")
        red = _reduce(part, 4 * Cc, B * T, 4 * Cc).reshape(4, Cc)
        st = hip_ops.col_stats(dF)   # (B, T, 2, 3): [..., 0, :] = column sums -> the bias gradient, summed in a fixed order
        db = _reduce(st, 3, B * st.shape[1], 2 * 3)
        return dy, red[:3].contiguous(), db


class LookupFn(torch.autograd.Function):
    """RayNetwork.extract_image_features with a gradient into the pyramid levels and — for a caller that asks — into the geometry
    (reference models/ray.py:64-87; in training the geometry is the noised data and carries none)."""

    @staticmethod
    def forward(ctx, geom, K, reparam_spec, *features):
        levels = hip_ops.to_channels_last_levels([f.detach() for f in features])
        rp = hip_ops.make_reparam(*reparam_spec)
        ctx.save_for_backward(geom, K, *levels)
        ctx.spec = reparam_spec
        return hip_ops.ray_lookup(geom, K, levels, rp)

    @staticmethod
    def backward(ctx, dout):
        geom, K, *levels = ctx.saved_tensors
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or any(ctx.needs_input_grad[3:])):
            return (None,) * (3 + len(levels))   # a frozen (or foreign, detached) conditioner: nothing to compute
        dout = _f(dout)
        lib = _lib.load()
        B, N, _ = geom.shape
        rp = hip_ops.make_reparam(*ctx.spec)
        pyr = hip_ops.make_pyramid(levels)
        dgeom = dK = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # a caller differentiates with respect to the input cloud (guidance) or the camera matrix: through taps, projection, reparam
            dgeom = _new(B, N, 3, like=dout) if ctx.needs_input_grad[0] else None
            T = lib.gecco_lookup_row_tiles(N)
            kpart = _new(B, T, 4, like=dout) if ctx.needs_input_grad[1] else None
            _lib.check(lib.gecco_ray_lookup_dgeom_f32(_ptr(_f(geom)), _ptr(_f(K.float())), C.byref(rp), C.byref(pyr), _ptr(dout), _ptr(dgeom),
                                                      _ptr(kpart), B, N, _stream()), "gecco_ray_lookup_dgeom_f32")
            if kpart is not None:   # (fx, cx, fy, cy) -> the (3, 3) entries they sit at
                k4 = kpart.sum(1)
                dK = torch.zeros(B, 3, 3, device=dout.device, dtype=torch.float32)
                dK[:, 0, 0], dK[:, 0, 2], dK[:, 1, 1], dK[:, 1, 2] = k4[:, 0], k4[:, 1], k4[:, 2], k4[:, 3]
                dK = dK.reshape(K.shape)
        if not any(ctx.needs_input_grad[3:]):
            return (dgeom, dK, None, *[None] * len(levels))
        nb = lib.gecco_ray_lookup_bwd_sorted_workspace_bytes(C.byref(pyr), B, N) if os.environ.get("GECCO_LOOKUP_BWD", "sorted") == "sorted" else 0
        if nb:   # sort + gather: no atomics, fixed summation order, every texel written
            grads = [torch.empty_like(f) for f in levels]   # (B, H, W, C)
            arr = (C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
            ws = hip_ops._ws(nb, geom.device)
            _lib.check(lib.gecco_ray_lookup_bwd_sorted_f32(_ptr(geom), None, _ptr(K), C.byref(rp), C.byref(pyr), _ptr(dout), arr, B, N,
                                                           C.c_void_p(ws.data_ptr()), nb, _stream()), "gecco_ray_lookup_bwd_sorted_f32")
        else:    # more than 4096 points per cloud: float atomics, like torch's grid_sampler backward
            grads = [torch.zeros_like(f) for f in levels]
            arr = (C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
            _lib.check(lib.gecco_ray_lookup_bwd_f32(_ptr(geom), None, _ptr(K), C.byref(rp), C.byref(pyr), _ptr(dout), arr, B, N,
                                                    _stream()), "gecco_ray_lookup_bwd_f32")
        # handed back NCHW-shaped (channels-last strides, no copy)
        return (dgeom, dK, None, *[g.permute(0, 3, 1, 2) for g in grads])


# ------------------------------------------------------------------------------------------- ConvNeXt conditioner
class GeluFn(torch.autograd.Function):
    """nn.GELU() (erf form) between the pointwise linears of a CNBlock."""

    @staticmethod
    def forward(ctx, u):
        u = _f(u)
        ctx.save_for_backward(u)
        y = torch.empty_like(u)
        _lib.check(_lib.load().gecco_gelu_f32(_ptr(u), _ptr(y), u.numel(), _stream()), "gecco_gelu_f32")
        return y

    @staticmethod
    def backward(ctx, dy):
        (u,) = ctx.saved_tensors
        dy = _f(dy)
        du = torch.empty_like(u)
        _lib.check(_lib.load().gecco_gelu_bwd_f32(_ptr(u), _ptr(dy), _ptr(du), u.numel(), _stream()), "gecco_gelu_bwd_f32")
        return du


def _cnx_ln_bwd(z: Tensor, dy: Tensor, ln_w: Tensor, eps: float, patch2: bool):
    """LayerNorm over C backward from its input z (B, H, W, C) -> dz, (d ln_w, d ln_b, column sums of dz)."""
    lib = _lib.load()
    B, H, W, Cc = z.shape
    nb = lib.gecco_convnext_ln_bwd_blocks(B, H, W, Cc)
    if nb <= 0:
        raise _lib.GeccoHipError("convnext LayerNorm backward: C must be 96, 192 or 384")
    dz, parts = torch.empty_like(z), _new(nb, 3 * Cc, like=z)
    _lib.check(lib.gecco_convnext_ln_bwd_f32(_ptr(z), _ptr(dy), _ptr(ln_w), _ptr(dz), _ptr(parts), B, H, W, Cc, eps, int(patch2),
                                             _stream()), "gecco_convnext_ln_bwd_f32")
    red = _reduce(parts, 3 * Cc, nb, 3 * Cc)
    return dz, red[:Cc], red[Cc:2 * Cc], red[2 * Cc:]


class CnxStemFn(torch.autograd.Function):
    """LayerNorm2d(Conv2d(3, C, k4, s4)(image)) -> (B, H/4, W/4, C)   (torchvision ConvNeXt features[0])."""

    @staticmethod
    def forward(ctx, img, w, b, ln_w, ln_b, eps):
        img = _f(img.float())
        B, _, H, W = img.shape
        Cc = w.shape[0]
        out = _new(B, H // 4, W // 4, Cc, like=img)
        z = torch.empty_like(out)
        _lib.check(_lib.load().gecco_convnext_stem_train_f32(_ptr(img), _ptr(_f(w)), _ptr(b), _ptr(ln_w), _ptr(ln_b), _ptr(out), _ptr(z),
                                                             B, H, W, Cc, eps, _stream()), "gecco_convnext_stem_train_f32")
        ctx.save_for_backward(img, z, ln_w, w)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, dy):
        img, z, ln_w, w = ctx.saved_tensors
        B, h, w_, Cc = z.shape
        dz, dg, dbl, dbias = _cnx_ln_bwd(z, _f(dy), ln_w, ctx.eps, False)
        dimg = None
        if ctx.needs_input_grad[0]:
            # the 4 x 4 patches do not overlap: d patch = dz W (a linear's dX product), then every value back to its pixel
            dp = _linear_dx(dz.view(1, B * h * w_, Cc), _f(w.reshape(Cc, 48)))
            dimg = dp.view(B, h, w_, 3, 4, 4).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, 4 * h, 4 * w_)
        patches = _new(B, h * w_, 48, like=z)
        _lib.check(_lib.load().gecco_convnext_im2col4_f32(_ptr(img), _ptr(patches), B, img.shape[2], img.shape[3], _stream()),
                   "gecco_convnext_im2col4_f32")
        dW = _linear_dw(dz.view(1, B * h * w_, Cc), patches.view(1, B * h * w_, 48)).reshape(Cc, 3, 4, 4)
        return dimg, dW, dbias, dg, dbl, None


class CnxDwLnFn(torch.autograd.Function):
    """LayerNorm(dwconv7x7(x) + b) on (B, H, W, C): the front half of a CNBlock (block.0 .. block.2)."""

    @staticmethod
    def forward(ctx, x, w, b, ln_w, ln_b, eps):
        x = _f(x)
        B, H, W, Cc = x.shape
        w_tap = w.reshape(Cc, 49).t().contiguous()   # tap-major (49, C)
        out, z = torch.empty_like(x), torch.empty_like(x)
        _lib.check(_lib.load().gecco_convnext_dwconv_ln_train_f32(_ptr(x), _ptr(w_tap), _ptr(b), _ptr(ln_w), _ptr(ln_b), _ptr(out),
                                                                  _ptr(z), B, H, W, Cc, eps, _stream()),
                   "gecco_convnext_dwconv_ln_train_f32")
        ctx.save_for_backward(x, z, w_tap, ln_w)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, dy):
        x, z, w_tap, ln_w = ctx.saved_tensors
        lib = _lib.load()
        B, H, W, Cc = x.shape
        dz, dg, dbl, dbias = _cnx_ln_bwd(z, _f(dy), ln_w, ctx.eps, False)
        dx = None
        if ctx.needs_input_grad[0]:   # the same convolution on the reversed taps
            dx = torch.empty_like(x)
            _lib.check(lib.gecco_convnext_dwconv_f32(_ptr(dz), _ptr(w_tap.flip(0).contiguous()), None, _ptr(dx), B, H, W, Cc, _stream()),
                       "gecco_convnext_dwconv_f32")
        nb = lib.gecco_convnext_dwconv_dw_blocks(B, H, W, Cc)
        parts = _new(nb, 49 * Cc, like=x)
        _lib.check(lib.gecco_convnext_dwconv_dw_f32(_ptr(x), _ptr(dz), _ptr(parts), B, H, W, Cc, _stream()), "gecco_convnext_dwconv_dw_f32")
        dW = _reduce(parts, 49 * Cc, nb, 49 * Cc).reshape(49, Cc).t().reshape(Cc, 1, 7, 7)
        return dx, dW, dbias, dg, dbl, None


class CnxBlockFn(torch.autograd.Function):
    """A whole CNBlock, x + layer_scale * Linear(4C -> C)(GELU(Linear(C -> 4C)(LayerNorm(dwconv7x7(x))))) (torchvision's CNBlock,
    stochastic depth off: models/feature_pyramid.py:55-59), as ONE Function.  Forward: tap-major re-layout, depthwise conv +
    LayerNorm (keeping the LayerNorm input), layer_scale folded into the second linear by one kernel, the first linear with
    GELU in its epilogue (pre-activation kept), the second with the skip as its epilogue.  Backward: GELU' as the epilogue of
    the second linear's dX product, the fold's backward in one kernel (dW2, db2, d layer_scale), LayerNorm backward, and the
    depthwise input gradient on reversed taps WITH the skip's gradient added in the same kernel — no autograd additions, no
    flipped or scaled copies made by torch."""

    @staticmethod
    def forward(ctx, x, dw_w, dw_b, ln_w, ln_b, W1, b1, W2, b2, ls, eps):
        x = _f(x)
        lib = _lib.load()
        B, H, W, Cc = x.shape
        rows = B * H * W
        w_tap = dw_w.reshape(Cc, 49).t().contiguous()   # tap-major (49, C)
        y, z = torch.empty_like(x), torch.empty_like(x)
        _lib.check(lib.gecco_convnext_dwconv_ln_train_f32(_ptr(x), _ptr(w_tap), _ptr(dw_b), _ptr(ln_w), _ptr(ln_b), _ptr(y), _ptr(z),
                                                          B, H, W, Cc, eps, _stream()), "gecco_convnext_dwconv_ln_train_f32")
        lsv = _f(ls.reshape(-1))
        w2f, b2f = torch.empty_like(W2), torch.empty_like(b2)
        _lib.check(lib.gecco_convnext_fold_scale_f32(_ptr(W2), _ptr(b2), _ptr(lsv), _ptr(w2f), _ptr(b2f), Cc, W2.shape[1], _stream()),
                   "gecco_convnext_fold_scale_f32")
        y3 = y.view(1, rows, Cc)
        N1 = W1.shape[0]
        ctx.prec = _lin_precision()   # under autocast(float16) the conditioner's pointwise linears run in fp16 like the reference's
        prec = _resolve(ctx.prec, rows, Cc, N1)
        if prec in ("fp32", "bf16x3", "fp16") and lib.gecco_linear_actbwd_ok(rows, Cc, N1, hip_ops.PRECISIONS[prec]):
            u, h = _new(1, rows, N1, like=x), _new(1, rows, N1, like=x)
            img = WEIGHT_IMAGES.lookup("n", W1, prec=prec) if prec in ("bf16x3", "fp16") and _image_ok(rows, Cc, N1, prec) else None
            Wp, ws = (None, img) if img is not None else (
                _f(W1), hip_ops._ws((N1 + 127) // 128 * 128 * Cc * 4, x.device) if prec != "fp32" else None)
            _lib.check(lib.gecco_linear_act_keep_f32(_ptr(y3), _ptr(Wp), _ptr(b1), None, 4, _ptr(u), _ptr(h), 1, rows, Cc, N1,
                                                     hip_ops.PRECISIONS[prec], C.c_void_p(ws.data_ptr()) if ws is not None else None,
                                                     _stream()), "gecco_linear_act_keep_f32")
        else:
            u = _linear_fwd(y3, W1, b1, None, False, ctx.prec)
            h = _act_forward(u, None, 4)
        out = _linear_fwd(h, w2f, b2f, x.view(1, rows, Cc), False, ctx.prec)
        ctx.save_for_backward(x, z, w_tap, ln_w, y, u, h, W1, W2, b2, lsv, w2f)
        ctx.eps, ctx.ls_shape = eps, ls.shape
        return out.view(B, H, W, Cc)

    @staticmethod
    def backward(ctx, dout):
        x, z, w_tap, ln_w, y, u, h, W1, W2, b2, lsv, w2f = ctx.saved_tensors
        lib = _lib.load()
        B, H, W, Cc = x.shape
        rows = B * H * W
        dout = _f(dout)
        d3 = dout.view(1, rows, Cc)
        prec = ctx.prec
        du, _ = _act_linear_dx(d3, u, h, None, w2f, 4, False, prec)
        dWp, dbp = _linear_dw(d3, h, want_db=True, prec=prec)
        dW2, db2, dls = torch.empty_like(W2), torch.empty_like(b2), torch.empty_like(lsv)
        _lib.check(lib.gecco_convnext_fold_scale_bwd_f32(_ptr(dWp), _ptr(dbp), _ptr(W2), _ptr(b2), _ptr(lsv), _ptr(dW2), _ptr(db2),
                                                         _ptr(dls), Cc, W2.shape[1], _stream()), "gecco_convnext_fold_scale_bwd_f32")
        dW1, db1 = _linear_dw(du, y.view(1, rows, Cc), want_db=True, leaf=W1, prec=prec)
        dy = _linear_dx(du, W1, prec=prec).view(B, H, W, Cc)
        dz, dg, dbl, dbias = _cnx_ln_bwd(z, dy, ln_w, ctx.eps, False)
        dx = None
        if ctx.needs_input_grad[0]:   # reversed taps, + the gradient that came through the skip, one launch
            dx = torch.empty_like(x)
            _lib.check(lib.gecco_convnext_dwconv_bwd_f32(_ptr(dz), _ptr(w_tap), _ptr(dout), _ptr(dx), B, H, W, Cc, _stream()),
                       "gecco_convnext_dwconv_bwd_f32")
        nb = lib.gecco_convnext_dwconv_dw_blocks(B, H, W, Cc)
        parts = _new(nb, 49 * Cc, like=x)
        _lib.check(lib.gecco_convnext_dwconv_dw_f32(_ptr(x), _ptr(dz), _ptr(parts), B, H, W, Cc, _stream()), "gecco_convnext_dwconv_dw_f32")
        dWd = _reduce(parts, 49 * Cc, nb, 49 * Cc).reshape(49, Cc).t().reshape(Cc, 1, 7, 7)
        return dx, dWd, dbias, dg, dbl, dW1, db1, dW2, db2, dls.reshape(ctx.ls_shape), None


class CnxLnPatchFn(torch.autograd.Function):
    """LayerNorm2d(x) gathered into the 2 x 2 stride-2 convolution's GEMM operand (B, H/2, W/2, (dy, dx, c))."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, eps):
        x = _f(x)
        B, H, W, Cc = x.shape
        out = _new(B, H // 2, W // 2, 4 * Cc, like=x)
        _lib.check(_lib.load().gecco_convnext_ln_patch2_f32(_ptr(x), _ptr(ln_w), _ptr(ln_b), _ptr(out), B, H, W, Cc, eps, _stream()),
                   "gecco_convnext_ln_patch2_f32")
        ctx.save_for_backward(x, ln_w)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, dp):
        x, ln_w = ctx.saved_tensors
        dx, dg, dbl, _ = _cnx_ln_bwd(x, _f(dp), ln_w, ctx.eps, True)
        return dx, dg, dbl, None


def convnext_pyramid(ext, image: Tensor) -> list[Tensor]:
    """ConvNeXtExtractor.forward WITH autograd (reference models/feature_pyramid.py:62-73 over torchvision's stages; the
    reference optimises the conditioner's parameters with the denoiser's, diffusion.py:210-211).  Channels-last like the
    inference path; returns NCHW-shaped views.  layer_scale and the 2 x 2 convolution's weight reach the GEMM as small
    re-laid-out parameter tensors (torch ops on parameters, differentiated by autograd)."""
    from .models.feature_pyramid import LN_EPS
    feats = []
    x = None
    for s, stage in enumerate(ext.stages):
        head, blocks = stage[0], stage[1]
        if s == 0:
            conv, ln = head[0], head[1]
            x = CnxStemFn.apply(image, conv.weight, conv.bias, ln.weight, ln.bias, LN_EPS)
        else:
            ln, conv = head[0], head[1]
            Cin, Cout = conv.in_channels, conv.out_channels
            patches = CnxLnPatchFn.apply(x, ln.weight, ln.bias, LN_EPS)
            Bq, hq, wq, _ = patches.shape
            wmat = conv.weight.permute(0, 2, 3, 1).reshape(Cout, 4 * Cin)
            x = LinearFn.apply(patches.view(1, Bq * hq * wq, 4 * Cin), wmat, conv.bias).view(Bq, hq, wq, Cout)
        Bq, hq, wq, Cc = x.shape
        rows = Bq * hq * wq
        fused = os.environ.get("GECCO_TRAIN_CNBLOCK", "1") != "0"
        for blk in blocks:
            dw, ln, pw1, pw2 = blk.block[0], blk.block[2], blk.block[3], blk.block[5]
            if fused:   # the whole block as one Function
                x = CnxBlockFn.apply(x, dw.weight, dw.bias, ln.weight, ln.bias, pw1.weight, pw1.bias, pw2.weight, pw2.bias,
                                     blk.layer_scale, LN_EPS)
                continue
            y = CnxDwLnFn.apply(x, dw.weight, dw.bias, ln.weight, ln.bias, LN_EPS)
            ls = blk.layer_scale.reshape(-1)
            x = LinearActLinearFn.apply(y.view(1, rows, Cc), pw1.weight, pw1.bias, None, pw2.weight * ls[:, None], pw2.bias * ls,
                                        x.view(1, rows, Cc), 4).view(Bq, hq, wq, Cc)
        feats.append(x.permute(0, 3, 1, 2))
    return feats


def _inproj_side_ok(W: Tensor, b: Tensor) -> bool:
    """The packed in_proj parameters are plain leaves without a gradient yet (`zero_grad(set_to_none=True)`): autograd will KEEP the joined
    gradient instead of adding it into an existing one, so the thirds' weight gradients and their join may live on the side stream."""
    return (os.environ.get("GECCO_TRAIN_INPROJ_SIDE", "1") != "0" and _side_enabled()
            and all(t.is_leaf and t._base is None and t.grad is None and t.requires_grad for t in (W, b)))


class InProjSplitFn(torch.autograd.Function):
    """nn.MultiheadAttention's packed in_proj (weight (3C, C), bias (3C)) as its query third and its key | value two thirds — views, as
    `W[:C]`, `W[C:]` are (models/set_transformer.py:112 -> torch's in_proj packing).  Autograd's own slice backward builds, per slice, a
    zero-filled full-size tensor, copies the slice's gradient into it and adds the two: ten 5-us launches per layer.  Here the two
    gradients are concatenated once (two launches)."""

    @staticmethod
    def forward(ctx, W, b, Cc):
        ctx.Cc = Cc
        ctx.set_materialize_grads(False)
        ctx.side_join = _inproj_side_ok(W, b)
        return W[:Cc], b[:Cc], W[Cc:], b[Cc:]

    @staticmethod
    def backward(ctx, gWq, gbq, gWkv, gbkv):
        Cc = ctx.Cc

        def join(a, b_, rows_a, rows_b):
            if a is None and b_ is None:
                return None
            ref = a if a is not None else b_
            a = ref.new_zeros(rows_a, *ref.shape[1:]) if a is None else a
            b_ = ref.new_zeros(rows_b, *ref.shape[1:]) if b_ is None else b_
            return torch.cat([a, b_], 0)
        if ctx.side_join and _SIDE["stream"] is not None and _side_enabled():
            # the thirds' gradients were formed on the weight-gradient side stream (`_linear_dw(side_ok=True)`): join them there, in its
            # order; the packed parameter has no .grad yet, so autograd only keeps the result — nothing on the main stream reads it before
            # the pass-end callback (or the data-parallel reducer's explicit sync) has ordered the side stream in front
            side, main = _SIDE["stream"], torch.cuda.current_stream()
            side.wait_stream(main)       # (a third that was formed on the main stream after all)
            with torch.cuda.stream(side):
                gW, gb = join(gWq, gWkv, Cc, 2 * Cc), join(gbq, gbkv, Cc, 2 * Cc)
            for t in (gWq, gbq, gWkv, gbkv):
                if t is not None:
                    t.record_stream(side)
            for t in (gW, gb):
                if t is not None:
                    t.record_stream(main)
            _SIDE["pending"] = True
            try:
                torch.autograd.Variable._execution_engine.queue_callback(sync_side_stream)
            except RuntimeError:
                sync_side_stream()
            return gW, gb, None
        return join(gWq, gWkv, Cc, 2 * Cc), join(gbq, gbkv, Cc, 2 * Cc), None


# ------------------------------------------------------------------------------------------- network composition
def adagn(mod, x, t, passthrough=False, stats=None):
    return AdaGNFn.apply(x, t, mod.scale.weight, mod.scale.bias, mod.bias.weight, mod.bias.bias, mod.gn.num_groups, mod.gn.eps,
                         passthrough, stats)


def mlp(mod, x, residual=None, want_stats=False):
    """nn.Sequential(Linear, act, Linear, ...) (reference models/mlp.py); `residual` is added by the last Linear's epilogue,
    which with `want_stats` also leaves the next norm's partial sums: returns (y, stats)."""
    from .models.activation import GaussianActivation
    mods = list(mod)   # Linear, act, Linear[, act, Linear ...] (models/mlp.py)
    n = len(mods)
    def kind_of(act):
        if isinstance(act, GaussianActivation):
            return (1 if act.normalized else 2), act.alpha
        if isinstance(act, torch.nn.ReLU):
            return 3, None
        if isinstance(act, torch.nn.Identity):
            return 0, None
        raise NotImplementedError(f"training on HIP: no backward for activation {type(act).__name__}")
    k0, a0 = kind_of(mods[1]) if n >= 3 else (0, None)
    if k0:   # Linear -> act -> Linear: one Function (the forward keeps u from the first GEMM's epilogue, the backward runs act'
        # as the epilogue of the second linear's dX product)
        x = LinearActLinearFn.apply(x, mods[0].weight, mods[0].bias, a0, mods[2].weight, mods[2].bias, residual if n == 3 else None,
                                    k0, want_stats and n == 3)
        i = 3
    else:
        x = LinearFn.apply(x, mods[0].weight, mods[0].bias, residual if n == 1 else None, want_stats and n == 1)
        i = 1
    while i + 1 < n:
        act, lin = mods[i], mods[i + 1]
        last = i + 2 >= n
        kind, alpha = kind_of(act)
        if kind:   # activation + the linear after it: one Function whose backward runs act' as the dX product's epilogue
            x = ActLinearFn.apply(x, alpha, lin.weight, lin.bias, residual if last else None, kind, want_stats and last)
        else:
            x = LinearFn.apply(x, lin.weight, lin.bias, residual if last else None, want_stats and last)
        i += 2
    return x   # (y, stats) with want_stats


def broadcasting_layer(layer, x, t, h=None, stats=None, want_stats=False):
    """BroadcastingLayer.forward with autograd (reference models/set_transformer.py:155-168, 92-117, 47-65).
    stats / want_stats: the GroupNorm partial sums of x handed from layer to layer (formed by the GEMM that wrote x);
    with want_stats the return value is (x, h, stats of x)."""
    from .models.activation import GaussianActivation
    bc = layer.broadcast
    H = bc.pool.num_heads
    Cc = x.shape[-1]
    R = x.shape[1]
    if os.environ.get("GECCO_TRAIN_INPROJ_SPLIT", "1") != "0":
        Wq, bq, Wkv, bkv = InProjSplitFn.apply(bc.unpool.in_proj_weight, bc.unpool.in_proj_bias, Cc)
        if _inproj_side_ok(bc.unpool.in_proj_weight, bc.unpool.in_proj_bias):
            Wq._gecco_side_ok = Wkv._gecco_side_ok = True    # (read by the Functions that differentiate with respect to them)
    else:   # plain slices: autograd's slice backward (A/B runs)
        W_, b_ = bc.unpool.in_proj_weight, bc.unpool.in_proj_bias
        Wq, bq, Wkv, bkv = W_[:Cc], b_[:Cc], W_[Cc:], b_[Cc:]
    n1 = layer.broadcast_norm
    fused_pair = h is None and _pro_ok(R, Cc, 3 * Cc) and Cc % 128 == 0
    if fused_pair:   # broadcast_norm as the prologue of kv_proj | q_proj: AdaGN(x) is never written
        io16 = _io16_ok(R, Cc, H, bc.pool.inducers.shape[2])
        KV, q, x = AdaGNPairFn.apply(x, t, n1.scale.weight, n1.scale.bias, n1.bias.weight, n1.bias.bias, n1.gn.num_groups, n1.gn.eps,
                                     stats, bc.pool.kv_proj.weight, Wq, bq, io16)
    else:
        y, x = adagn(n1, x, t, passthrough=True, stats=stats)
    if h is None:
        if not fused_pair:
            KV, q = LinearPairFn.apply(y, bc.pool.kv_proj.weight, None, Wq, bq)
        merged = PoolAttnFn.apply(KV, bc.pool.inducers, H)
        h = LinearFn.apply(merged, bc.pool.out_proj.weight, None)
        h = adagn(bc.norm_1, h, t)
        h = mlp(bc.mlp, h)
        h = adagn(bc.norm_2, h, t)
    else:
        q = LinearFn.apply(y, Wq, bq)
    kvh = LinearFn.apply(h, Wkv, bkv)
    attn = UnpoolAttnFn.apply(q, kvh, H)
    x, st = LinearFn.apply(attn, bc.unpool.out_proj.weight, bc.unpool.out_proj.bias, x, True)   # x + out_proj(attn)
    n2, mods = layer.mlp_norm, list(layer.mlp)
    if len(mods) == 3 and isinstance(mods[1], (GaussianActivation, torch.nn.ReLU)) and _pro_ok(R, Cc, mods[0].weight.shape[0]):
        # x + mlp(mlp_norm(x)) as one Function: the norm is the first GEMM's prologue, the skip the second's epilogue
        act = mods[1]
        kind, alpha = ((1 if act.normalized else 2), act.alpha) if isinstance(act, GaussianActivation) else (3, None)
        out = AdaGNMlpFn.apply(x, t, n2.scale.weight, n2.scale.bias, n2.bias.weight, n2.bias.bias, n2.gn.num_groups, n2.gn.eps, st,
                               mods[0].weight, mods[0].bias, alpha, mods[2].weight, mods[2].bias, kind, want_stats)
        return (out[0], h, out[1]) if want_stats else (out, h)
    y, x = adagn(n2, x, t, passthrough=True, stats=st)
    if want_stats:
        x, st = mlp(layer.mlp, y, residual=x, want_stats=True)                            # x + mlp(mlp_norm(x))
        return x, h, st
    return mlp(layer.mlp, y, residual=x), h


def set_transformer(st, feats, t, return_h=False, hs=None):
    hs = [None] * len(st.layers) if hs is None else hs
    stored = []
    stats = None
    for layer, h in zip(st.layers, hs):
        feats, h, stats = broadcasting_layer(layer, feats, t, h, stats, True)
        stored.append(h)
    return feats, (stored if return_h else None)


def group_norm(x, G, eps):
    return AdaGNFn.apply(x, None, None, None, None, None, G, eps)


def ray_network(net, geometry, t, K, features, do_cache=False, cache=None):
    """RayNetwork.forward with autograd (reference models/ray.py:89-120)."""
    g = _f(geometry.float())
    xyz = LiftFn.apply(g, net.xyz_embed.weight, net.xyz_embed.bias)
    raw = LookupFn.apply(g, _f(K.float()), net.reparam.lookup_spec(), *features)
    gn, lin = net.img_feature_proj[0], net.img_feature_proj[1]
    feats = LinearFn.apply(group_norm(raw, gn.num_groups, gn.eps), lin.weight, lin.bias, xyz)   # xyz + img_feature_proj(raw)
    feats, out_cache = set_transformer(net.backbone, feats, t, do_cache, cache)
    gn2, lin2 = net.output_proj[0], net.output_proj[1]
    return Linear3Fn.apply(group_norm(feats, gn2.num_groups, gn2.eps), lin2.weight, lin2.bias), out_cache


def ray_network_edm(net, x, sigma, sigma_data, K, features, do_cache=False, cache=None):
    """EDMPrecond(RayNetwork).forward with autograd (reference diffusion.py:46-57)."""
    sigma = sigma.reshape(-1, 1, 1).float()
    sd = float(sigma_data)
    c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
    c_out = sigma * sd / (sigma ** 2 + sd ** 2).sqrt()
    c_in = 1 / (sd ** 2 + sigma ** 2).sqrt()
    c_noise = sigma.log() / 4
    F_x, out_cache = ray_network(net, c_in * x, c_noise, K, features, do_cache, cache)
    den = c_skip * x + c_out * F_x
    return (den, out_cache) if do_cache else den


def linear_lift_edm(net, x, sigma, sigma_data, do_cache=False, cache=None):
    """EDMPrecond(LinearLift).forward with autograd (reference diffusion.py:46-57, linear_lift.py:44-46)."""
    sigma = sigma.reshape(-1, 1, 1).float()
    sd = float(sigma_data)
    c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
    c_out = sigma * sd / (sigma ** 2 + sd ** 2).sqrt()
    c_in = 1 / (sd ** 2 + sigma ** 2).sqrt()
    c_noise = sigma.log() / 4
    feats = LiftFn.apply(c_in * x, net.lift.weight, net.lift.bias)
    feats, out_cache = set_transformer(net.inner, feats, c_noise, do_cache, cache)
    F_x = LowerFn.apply(feats, net.lower[1].weight, net.lower[1].bias, net.lower[0].eps)
    den = c_skip * x + c_out * F_x
    return (den, out_cache) if do_cache else den
