"""Optimizer of the training path: Adam fused with the EMA shadow weights, one HIP launch per step.

Reference: `Diffusion.configure_optimizers` returns `torch.optim.Adam(lr=1e-4)` (diffusion.py:210-211); `EMACallback`
wraps it in `EMAOptimizer` (ema.py:61-75), whose `step()` runs the inner optimizer and then `ema_update`
(ema.py:187-194, 273-325).  `FusedAdamEMA` is both at once:

* every parameter, its gradient, `exp_avg`, `exp_avg_sq` and its EMA shadow live as views of five flat, 16-byte aligned
  fp32 buffers; a step is ONE `gecco_adam_ema_step_f32` launch over them (36 bytes per parameter, HBM-bound);
* `p.grad` is a view of the flat gradient buffer, so autograd accumulates in place, a data-parallel all-reduce works
  on slices of that buffer (`gecco_amd.distributed.BucketedGradAllReducer`) and the 1 / world_size of the gradient
  mean folds into the kernel's read of g — no gather / scatter of gradients anywhere;
* the state-dict wire format is the reference's: `state_dict()` returns `EMAOptimizer.state_dict()`'s dict
  ({"opt": <torch.optim.Adam state_dict>, "ema": tuple of tensors, "current_step", "decay", "every_n_steps"},
  ema.py:369-388) and `load_state_dict` accepts it (or a bare Adam state dict), so Lightning checkpoints written by the
  reference resume here and vice versa (`gecco_amd/checkpoint.py`).

There is no CPU fallback: parameters must live on the HIP device when `step()` runs.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Any, Iterable

import torch
from torch import Tensor

from . import _lib


def _align4(n: int) -> int:
    return (n + 3) // 4 * 4


class FusedAdamEMA(torch.optim.Optimizer):
    """torch.optim.Adam (single param-group hyper-parameters, no amsgrad / maximize) + EMA of the parameters.

    `ema_decay=None` disables the shadow weights (plain fused Adam).  The EMA is updated on steps where
    `current_step % every_n_steps == 0`, counted like `EMAOptimizer` (ema.py:296-299)."""

    def __init__(self, params: Iterable[Tensor] | Iterable[dict], lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, ema_decay: float | None = 0.9999, every_n_steps: int = 1,
                 current_step: int = 0, missing_grad: str = "raise", amp_on_device: bool = False):
        if ema_decay is not None and not 0.0 <= ema_decay <= 1.0:
            raise ValueError("EMA decay value must be between 0 and 1")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False)
        super().__init__(params, defaults)
        self.decay = ema_decay
        self.every_n_steps = every_n_steps
        self.current_step = current_step          # EMAOptimizer's counter (ema.py:288)
        self.in_saving_ema_model_context = False  # ema.py:378-383
        self.save_original_optimizer_state = False
        self._flat: dict[str, Tensor] | None = None
        self._spans: list[tuple[Tensor, int, int]] = []   # (param, offset, numel) in param-group order
        self._span_of: dict[int, tuple[int, int]] | None = None
        self._adam_step = 0
        self._grad_mult = 1.0                     # `grad_scale`: set by a summing gradient all-reduce to 1 / world_size
        self._amp_scale: Tensor | None = None     # torch.amp.GradScaler's scale tensor while scaler.step(self) runs (see grad_scale)
        self._amp_skipped: Tensor | None = None   # device counter of the steps a GradScaler's found_inf skipped
        # The GradScaler protocol below is OPT-IN, per instance.  With the attribute set, Lightning's MixedPrecision plugin (the
        # reference's trainer) treats the optimizer as unscaling internally: it skips `scaler.unscale_` and its `clip_gradients`
        # refuses a clip value — and both shipped configs combine precision="16-mixed" with gradient_clip_val=1.0
        # (example_configs/*.py).  The default (False) is therefore the host path every trainer knows: unscale -> clip -> step,
        # found_inf read on the host by GradScaler.  amp_on_device=True: the scale and found_inf stay on the device (one launch, no
        # host read; gradient clipping is then the caller's business, on unscaled gradients it does not have).
        if amp_on_device:
            self._step_supports_amp_scaling = True
        self.amp_on_device = bool(amp_on_device)
        # A parameter whose .grad is None at step() (zero_grad(set_to_none=True) and no gradient arrived): torch.optim.Adam skips it;
        # the one-launch update cannot.  "raise" (default) refuses the step, "zero" updates it with a zero gradient (moments decay,
        # the step count advances).  With zero_grad() (views of the flat buffer) a missing gradient is indistinguishable from zeros.
        if missing_grad not in ("raise", "zero"):
            raise ValueError("missing_grad must be 'raise' or 'zero'")
        self.missing_grad = missing_grad
        self._missing_grad = 0
        self._missing_ids: set[int] = set()   # id(p) of the trainable parameters without a gradient this step

    # ------------------------------------------------------------------------------------------ GradScaler protocol
    # torch.amp.GradScaler.step(optimizer) (the reference's `precision="16-mixed"` trainer: example_configs/*.py) has two paths.
    # For a plain optimizer it unscales every gradient and then READS found_inf BACK ON THE HOST (`.item()`) to decide whether to
    # call step(): the host waits for the whole backward pass every step, and the ~15 ms of Python / launch work of the next step no
    # longer overlap the device.  An optimizer that declares `_step_supports_amp_scaling` receives the scale and found_inf
    # tensors instead (`optimizer.grad_scale = scale; optimizer.found_inf = found_inf; optimizer.step(); del ...`) and decides on
    # the device (gecco_adam_ema_step_amp_f32): a skipped step writes nothing — parameters, moments, EMA — exactly like the
    # skipped optimizer.step() of the host path, and Adam's step count does not advance (counted on the device: `_amp_skipped`).
    # `grad_scale` therefore has two faces: a float is this optimizer's own multiplier (1 / world size), a tensor is the scaler's.
    # (Instance attribute, set by amp_on_device=True: see __init__.)  With every_n_steps > 1 an EMA update that falls on a skipped
    # step is lost, not deferred, and `current_step` counts skipped steps too (the host cannot know): use every_n_steps == 1 with it.

    @property
    def grad_scale(self):
        return self._grad_mult

    @grad_scale.setter
    def grad_scale(self, value) -> None:
        if isinstance(value, Tensor) or value is None:   # GradScaler: its scale (None: the gradients are already unscaled)
            self._amp_scale = value
        else:
            self._grad_mult = float(value)

    @grad_scale.deleter
    def grad_scale(self) -> None:
        self._amp_scale = None

    @property
    def adam_steps_taken(self) -> int:
        """Adam's step count: update launches minus the steps a GradScaler skipped (reads the device counter: synchronises)."""
        return self._adam_step - (int(self._amp_skipped.item()) if self._amp_skipped is not None else 0)

    # ------------------------------------------------------------------------------------------ flat storage
    def all_parameters(self) -> list[Tensor]:
        return [p for g in self.param_groups for p in g["params"]]

    def _build(self) -> None:
        ps = self.all_parameters()
        if not ps:
            raise ValueError("FusedAdamEMA: no parameters")
        dev = ps[0].device
        if dev.type != "cuda":
            raise _lib.GeccoHipError("FusedAdamEMA needs parameters on the HIP device (no CPU fallback)")
        if any(p.dtype != torch.float32 or p.device != dev for p in ps):
            raise _lib.GeccoHipError("FusedAdamEMA: all parameters must be fp32 on one device")
        off, spans = 0, []
        for p in ps:
            spans.append((p, off, p.numel()))
            off += _align4(p.numel())             # every view starts 16-byte aligned
        n = off
        old = self._flat
        flat = {k: torch.zeros(n, dtype=torch.float32, device=dev) for k in ("p", "g", "m", "v")}
        flat["ema"] = torch.zeros(n, dtype=torch.float32, device=dev) if self.decay is not None else None
        with torch.no_grad():
            for i, (p, o, k) in enumerate(spans):
                flat["p"][o:o + k].copy_(p.detach().reshape(-1))
                if p.grad is not None:
                    flat["g"][o:o + k].copy_(p.grad.reshape(-1))
                if old is not None and i < len(self._spans) and self._spans[i][0] is p:   # keep state across a rebuild
                    oo = self._spans[i][1]
                    for key in ("m", "v", "ema"):
                        if flat[key] is not None and old.get(key) is not None:
                            flat[key][o:o + k].copy_(old[key][oo:oo + k])
                elif flat["ema"] is not None:
                    flat["ema"][o:o + k].copy_(p.detach().reshape(-1))   # EMA starts as a copy (ema.py:283-291)
                p.data = flat["p"][o:o + k].view(p.shape)
                p.grad = flat["g"][o:o + k].view(p.shape)
        self._flat, self._spans = flat, spans
        self._span_of = None
        from .autograd import WEIGHT_IMAGES
        WEIGHT_IMAGES.invalidate()   # the parameters moved into the flat buffer
        from . import hip_ops
        hip_ops.weights_changed()

    def _ensure(self) -> None:
        """(Re)build the flat views when parameters were added, moved or re-allocated since the last step."""
        ps = self.all_parameters()
        ok = self._flat is not None and len(ps) == len(self._spans)
        if ok:
            base = self._flat["p"].data_ptr()
            for (p, o, k), q in zip(self._spans, ps):
                if p is not q or q.data_ptr() != base + 4 * o:
                    ok = False
                    break
        if not ok:
            self._build()

    def flat_grad(self) -> Tensor:
        """The flat gradient buffer every p.grad is a view of (what a data-parallel all-reduce operates on)."""
        self._ensure()
        return self._flat["g"]

    def spans(self) -> list[tuple[Tensor, int, int]]:
        self._ensure()
        return list(self._spans)

    def view_of(self, key: str, i: int) -> Tensor:
        p, o, k = self._spans[i]
        return self._flat[key][o:o + k].view(p.shape)

    @property
    def ema_params(self) -> tuple[Tensor, ...]:
        self._ensure()
        if self._flat["ema"] is None:
            return ()
        return tuple(self.view_of("ema", i) for i in range(len(self._spans)))

    # ------------------------------------------------------------------------------------------ step
    def zero_grad(self, set_to_none: bool = False) -> None:
        """set_to_none=False (default): the flat gradient buffer is cleared and every p.grad stays a view of it — autograd
        accumulates in place (one small add per parameter per backward).
        set_to_none=True: p.grad = None, so autograd HANDS OVER the gradient tensors its backward functions produced instead
        of adding them to zeros (no per-parameter kernel); `step()` — or the data-parallel reducer, bucket by bucket as the
        backward completes them — gathers them into the flat buffer with one multi-tensor copy.  Same values either way."""
        self._ensure()
        if set_to_none:
            for p, _, _ in self._spans:
                p.grad = None
            return
        self._flat["g"].zero_()
        for p, o, k in self._spans:
            if p.grad is None or p.grad.data_ptr() != self._flat["g"].data_ptr() + 4 * o:
                p.grad = self._flat["g"][o:o + k].view(p.shape)

    @torch.no_grad()
    def gather_grads(self, params: Iterable[Tensor] | None = None) -> None:
        """Gradients that live in tensors of their own (zero_grad(set_to_none=True), or a caller that reset p.grad) are copied
        into their slices of the flat buffer — one multi-tensor copy — and p.grad becomes the view again; a parameter without a
        gradient gets zeros.  `params`: a subset (a reducer bucket); default all."""
        from .autograd import sync_side_stream
        sync_side_stream()   # weight gradients issued on the side stream (autograd._linear_dw), if any are still unordered
        if self._span_of is None or len(self._span_of) != len(self._spans):
            self._span_of = {id(p): (o, k) for p, o, k in self._spans}
        gb = self._flat["g"].data_ptr()
        src, dst = [], []
        for p in (params if params is not None else (q for q, _, _ in self._spans)):
            o, k = self._span_of[id(p)]
            gr = p.grad
            if gr is not None and gr.data_ptr() == gb + 4 * o:
                continue                       # still (or already) our view
            v = self._flat["g"][o:o + k].view(p.shape)
            if gr is None:
                v.zero_()
                # a FROZEN parameter (requires_grad False) never has a gradient: with its zero moments the update leaves it
                # where it is (step() refuses weight decay in that case), as torch.optim.Adam's skip does — not "missing"
                if p.requires_grad and id(p) not in self._missing_ids:
                    self._missing_ids.add(id(p))
                    self._missing_grad += 1
            else:
                src.append(gr)
                dst.append(v)
            p.grad = v
        if src:
            torch._foreach_copy_(dst, src)

    def _gather_foreign_grads(self) -> None:
        self.gather_grads()

    def _should_update_at_step(self) -> bool:
        return self.decay is not None and self.current_step % self.every_n_steps == 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._ensure()
        self._gather_foreign_grads()
        if self._missing_grad and self.missing_grad != "zero":
            n, self._missing_grad = self._missing_grad, 0
            self._missing_ids.clear()
            self.__dict__.pop("found_inf", None)   # (GradScaler.step does not clean up behind a step that raises)
            self._amp_scale = None
            raise RuntimeError(
                f"FusedAdamEMA.step(): {n} trainable parameter(s) received no gradient this step.  torch.optim.Adam (and the "
                "reference's EMAOptimizer around it) SKIPS such parameters; this optimizer's one-launch update over the flat buffers "
                "would instead decay their moments, move them by stale momentum and advance their step count.  Leave them out of "
                "the optimizer's parameter list (or set requires_grad_(False): a frozen parameter with zero moments does not move), "
                "or pass missing_grad='zero' to accept a zero gradient.")
        self._missing_grad = 0
        self._missing_ids.clear()
        g = self.param_groups[0]
        if float(g["weight_decay"]) != 0.0 and any(not p.requires_grad for p, _, _ in self._spans):
            raise NotImplementedError("FusedAdamEMA: weight_decay with frozen parameters in the parameter list (the one-launch update "
                                      "would decay them; torch.optim.Adam skips them): leave frozen parameters out of the list")
        if len(self.param_groups) > 1 and any(
                (h["lr"], h["betas"], h["eps"], h["weight_decay"]) != (g["lr"], g["betas"], g["eps"], g["weight_decay"])
                for h in self.param_groups[1:]):
            raise NotImplementedError("FusedAdamEMA: one set of hyper-parameters for all param groups")
        if g.get("amsgrad") or g.get("maximize"):
            raise NotImplementedError("FusedAdamEMA: amsgrad / maximize are not supported")
        self._adam_step += 1
        try:
            self.launch(self._adam_step, self._should_update_at_step())
        except BaseException:
            # GradScaler.step deletes these itself when step() returns normally (and fails if they are gone by then): only a step
            # that raised must not leave a stale scale / found_inf for a later plain step()
            self.__dict__.pop("found_inf", None)
            self._amp_scale = None
            raise
        from .autograd import WEIGHT_IMAGES
        WEIGHT_IMAGES.invalidate()   # the kernel updates the weights through raw pointers: no version counter moves
        from . import hip_ops
        hip_ops.weights_changed()    # (nor does an inference plan inside hip_ops.frozen_weights() see it otherwise)
        self.current_step += 1
        return loss

    def launch(self, adam_step: int, do_ema: bool) -> None:
        """The one kernel of a step (gecco_adam_ema_step_f32) on the current stream, no bookkeeping."""
        f, g = self._flat, self.param_groups[0]
        a = _lib.GeccoAdamEma(f["p"].data_ptr(), f["g"].data_ptr(), f["m"].data_ptr(), f["v"].data_ptr(),
                              f["ema"].data_ptr() if f["ema"] is not None else None, f["p"].numel(), float(g["lr"]),
                              float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                              float(self.decay if self.decay is not None else 0.0), float(self.grad_scale), adam_step,
                              int(do_ema))
        found_inf = self.__dict__.get("found_inf")   # set (and deleted again) by GradScaler.step around step()
        if found_inf is None:
            _lib.check(_lib.load().gecco_adam_ema_step_f32(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "gecco_adam_ema_step_f32")
            return
        if self._amp_skipped is None:
            self._amp_skipped = torch.zeros(1, dtype=torch.int32, device=f["p"].device)
        found_inf = found_inf.to(device=f["p"].device, dtype=torch.float32)
        scale = self._amp_scale.to(device=f["p"].device, dtype=torch.float32) if self._amp_scale is not None else None
        _lib.check(_lib.load().gecco_adam_ema_step_amp_f32(C.byref(a), C.c_void_p(scale.data_ptr()) if scale is not None else None,
                                                           C.c_void_p(found_inf.data_ptr()), C.c_void_p(self._amp_skipped.data_ptr()),
                                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "gecco_adam_ema_step_amp_f32")

    # ------------------------------------------------------------------------------------------ EMA weight swap
    def join(self) -> None:   # EMAOptimizer API (its update runs on a side stream / thread; ours is in-stream)
        pass

    def switch_main_parameter_weights(self, saving_ema_model: bool = False) -> None:
        """In-place swap of the parameters with their EMA shadows (ema.py:335-339)."""
        self._ensure()
        if self._flat["ema"] is None:
            raise RuntimeError("EMA is disabled (ema_decay=None)")
        self.in_saving_ema_model_context = saving_ema_model
        tmp = self._flat["p"].clone()
        self._flat["p"].copy_(self._flat["ema"])
        self._flat["ema"].copy_(tmp)
        from .autograd import WEIGHT_IMAGES
        WEIGHT_IMAGES.invalidate()   # the parameters are views of the flat buffer: their version counters did not move
        from . import hip_ops
        hip_ops.weights_changed()    # an inference plan inside frozen_weights() (or a frozen graph's next capture) must rebuild its
                                     # streamed weight images: otherwise it would mix pre-swap images with post-swap biases

    @contextlib.contextmanager
    def swap_ema_weights(self, enabled: bool = True):
        if enabled:
            self.switch_main_parameter_weights()
        try:
            yield
        finally:
            if enabled:
                self.switch_main_parameter_weights()

    # ------------------------------------------------------------------------------------------ wire format
    def _adam_state_dict(self) -> dict[str, Any]:
        """What torch.optim.Adam.state_dict() returns for the same parameters (packed ids, per-parameter state)."""
        self._ensure()
        state, idx, groups = {}, 0, []
        steps_taken = self.adam_steps_taken
        for g in self.param_groups:
            ids = []
            for _ in g["params"]:
                if self._adam_step > 0:
                    state[idx] = {"step": torch.tensor(float(steps_taken)),
                                  "exp_avg": self.view_of("m", idx).clone(), "exp_avg_sq": self.view_of("v", idx).clone()}
                ids.append(idx)
                idx += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        return {"state": state, "param_groups": groups}

    def state_dict(self) -> dict[str, Any]:
        opt = self._adam_state_dict()
        if self.decay is None or self.save_original_optimizer_state:
            return opt
        # in the context of saving an EMA model the EMA weights sit in the modules' own weights (ema.py:378-383)
        ema = tuple(p.detach().clone() for p in self.all_parameters()) if self.in_saving_ema_model_context \
            else tuple(t.clone() for t in self.ema_params)
        # EMAOptimizer.step is not called on a step the GradScaler skips (ema.py:288 counts the steps taken): with the scaler's protocol
        # on the device the host counter ran on, so the saved value leaves the skipped steps out, as Adam's own step count does
        skipped = int(self._amp_skipped.item()) if self._amp_skipped is not None else 0
        return {"opt": opt, "ema": ema, "current_step": self.current_step - skipped, "decay": self.decay,
                "every_n_steps": self.every_n_steps}

    @torch.no_grad()
    def load_state_dict(self, state_dict: dict[str, Any]) -> None:
        self._ensure()
        opt = state_dict["opt"] if "opt" in state_dict else state_dict
        n = len(self._spans)
        ids = [i for g in opt["param_groups"] for i in g["params"]]
        if len(ids) != n:
            raise ValueError(f"loaded state dict has {len(ids)} parameters, the optimizer {n}")
        for g, sg in zip(self.param_groups, opt["param_groups"]):
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in sg:
                    g[k] = tuple(sg[k]) if k == "betas" else sg[k]
        steps = set()
        self._flat["m"].zero_()
        self._flat["v"].zero_()
        for pos, i in enumerate(ids):
            st = opt["state"].get(i)
            if st is None:
                continue
            self.view_of("m", pos).copy_(st["exp_avg"])
            self.view_of("v", pos).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise NotImplementedError("FusedAdamEMA: parameters with different Adam step counts")
        self._adam_step = steps.pop() if steps else 0
        self._amp_skipped = None
        if "opt" in state_dict:
            if self._flat["ema"] is None:
                self.decay = state_dict["decay"]
                self._build_ema_buffer()
            ema = state_dict["ema"]
            if len(ema) != n:
                raise ValueError(f"loaded EMA has {len(ema)} tensors, the optimizer {n} parameters")
            for pos, t in enumerate(ema):
                self.view_of("ema", pos).copy_(t)
            self.current_step = state_dict["current_step"]
            self.decay = state_dict["decay"]
            self.every_n_steps = state_dict["every_n_steps"]

    def _build_ema_buffer(self) -> None:
        self._flat["ema"] = self._flat["p"].clone()
