"""EDM diffusion logic on the MI355X HIP path (API of reference diffusion.py:22-470): preconditioning, noise
schedules, loss, the stochastic Heun sampler and the inducer-cached upsampler.

Differences from the reference are in *how*, not *what*:
* `EDMPrecond` hands the whole evaluation (c_in scaling, lift / projective lookup, set transformer, lower, c_skip /
  c_out combine) to one C call when the wrapped model offers `fused_edm` (LinearLift, RayNetwork);
* the sampler keeps its fp64 state in HIP kernels driven by a device-resident schedule table and a device step
  counter, so one captured hipGraph per step is replayed for the whole trajectory and the host never reads a
  device scalar (the reference compares `t_cur` on the host every step, diffusion.py:318-322);
* all noise of a trajectory is drawn up front (or injected with `noise=` for bit-reproducible parity tests).
"""
from __future__ import annotations

import ctypes as C
import functools
import math
from typing import Any, Sequence

import torch
from torch import Tensor, nn

from . import _lib, hip_ops
from ._grad import GeccoTrainingNotSupported
from .reparam import NoReparam, Reparam
from .structs import Context3d, Example


def _frozen_weights(fn):
    """A sampling call evaluates the denoiser hundreds of times on weights that cannot change while it runs: its evaluations share
    one build of the weight images per workspace (hip_ops.frozen_weights) instead of rebuilding them every time."""
    @functools.wraps(fn)
    def run(*a, **k):
        with hip_ops.frozen_weights():
            return fn(*a, **k)
    return run

try:  # Lightning is the reference's training harness; optional here (not needed for sampling)
    import lightning.pytorch as pl
    _Base = pl.LightningModule
except Exception:  # pragma: no cover - depends on the environment
    class _Base(nn.Module):
        def log(self, *args, **kwargs):
            pass


def ones(n: int):
    return (1,) * n


class EDMPrecond(nn.Module):
    """Karras et al. preconditioning: D(x, sigma) = c_skip x + c_out F(c_in x, ln(sigma)/4)."""

    def __init__(self, model: nn.Module, sigma_data=1.0):
        super().__init__()
        self.model = model
        self.sigma_data = sigma_data

    def forward(self, x: Tensor, sigma: Tensor, raw_context: Any, post_context: Any, do_cache: bool = False,
                cache: list[Tensor] | None = None, out: Tensor | None = None):
        sigma = sigma.reshape(-1)
        if hasattr(self.model, "fused_edm"):
            res = self.model.fused_edm(x, sigma, raw_context, post_context, do_cache, cache, float(self.sigma_data), out=out)
            return res  # Tensor, or (Tensor, cache) iff do_cache — as the reference
        raise NotImplementedError(
            f"EDMPrecond on HIP wraps LinearLift or RayNetwork (got {type(self.model).__name__}); there is no eager fallback")


class LogNormalSchedule(nn.Module):
    """sigma = exp(N(mean, std)) (Karras et al.).  (The reference's forward reads undefined attributes,
    diffusion.py:84; here it uses its own mean/std.)"""

    def __init__(self, sigma_max: float, mean=-1.2, std=1.2):
        super().__init__()
        self.sigma_max = sigma_max
        self.mean = mean
        self.std = std

    def extra_repr(self) -> str:
        return f"sigma_max={self.sigma_max}, mean={self.mean}, std={self.std}"

    def forward(self, data: Tensor) -> Tensor:
        rnd = torch.randn([data.shape[0], *ones(data.ndim - 1)], device=data.device)
        return (rnd * self.std + self.mean).exp()


class LogUniformSchedule(nn.Module):
    """Stratified log-uniform sigma in [min, max] (reference diffusion.py:87-115)."""

    def __init__(self, max: float, min: float = 0.002, low_discrepancy: bool = True):
        super().__init__()
        self.sigma_min = min
        self.sigma_max = max
        self.log_sigma_min = math.log(min)
        self.log_sigma_max = math.log(max)
        self.low_discrepancy = low_discrepancy

    def extra_repr(self) -> str:
        return f"sigma_min={self.sigma_min}, sigma_max={self.sigma_max}, low_discrepancy={self.low_discrepancy}"

    def forward(self, data: Tensor) -> Tensor:
        u = torch.rand(data.shape[0], device=data.device)
        if self.low_discrepancy:
            div = 1 / data.shape[0]
            u = div * u + div * torch.arange(data.shape[0], device=data.device)
        sigma = (u * (self.log_sigma_max - self.log_sigma_min) + self.log_sigma_min).exp()
        return sigma.reshape(-1, *ones(data.ndim - 1))


class EDMLoss(nn.Module):
    """Weighted denoising loss (reference diffusion.py:118-143).  With grad enabled the denoiser runs on the unfused
    HIP training path (gecco_amd/autograd.py), so `loss.backward()` produces every parameter gradient in HIP."""

    def __init__(self, schedule: nn.Module, sigma_data: float = 1.0, loss_scale: float = 100.0):
        super().__init__()
        self.schedule = schedule
        self.sigma_data = sigma_data
        self.loss_scale = loss_scale

    def extra_repr(self) -> str:
        return f"sigma_data={self.sigma_data}, loss_scale={self.loss_scale}"

    def forward(self, net: "Diffusion", examples: Tensor, context: Context3d) -> Tensor:
        ex_diff = net.reparam.data_to_diffusion(examples, context)
        sigma = self.schedule(ex_diff)
        weight = (sigma ** 2 + self.sigma_data ** 2) / ((sigma * self.sigma_data) ** 2)
        n = torch.randn_like(ex_diff) * sigma
        D_yn = net(ex_diff + n, sigma, context)
        return (self.loss_scale * weight * ((D_yn - ex_diff) ** 2)).mean()


class Conditioner(nn.Module):
    def forward(self, raw_context):
        raise NotImplementedError()


class IdleConditioner(Conditioner):
    def forward(self, raw_context: Context3d | None) -> None:
        del raw_context
        return None


# ---------------------------------------------------------------------------------------------- sampler engine
def karras_t_steps(num_steps: int, sigma_max: float, sigma_min: float, rho: float) -> Tensor:
    """fp64 Karras schedule with t_N = 0 (reference diffusion.py:253-269), on the host."""
    i = torch.arange(num_steps, dtype=torch.float64)
    t = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros_like(t[:1])])


def build_schedule_table(t_steps: Tensor, num_steps: int, S_churn, S_min, S_max, S_noise) -> Tensor:
    """Host table (num_steps, GECCO_SCHED_COLS) of fp64: {t_cur, t_hat, t_next, churn, redo, 0, 0, 0}."""
    tab = torch.zeros(num_steps, 8, dtype=torch.float64)
    for i in range(num_steps):
        t_cur, t_next = t_steps[i], t_steps[i + 1]
        gamma = min(S_churn / num_steps, math.sqrt(2.0) - 1) if S_min <= float(t_cur) <= S_max else 0
        t_hat = t_cur + gamma * t_cur
        tab[i, 0], tab[i, 1], tab[i, 2] = t_cur, t_hat, t_next
        tab[i, 3] = (t_hat ** 2 - t_cur ** 2).sqrt() * S_noise
        tab[i, 4] = (t_cur ** 2 - t_next ** 2).sqrt()
    return tab


def _vp(t: Tensor | None):
    return C.c_void_p(0 if t is None else t.data_ptr())


class _SamplerState:
    """Device buffers + kernel launches of one trajectory over `shape` = (B, n, 3)."""

    def __init__(self, shape, device, sched_host: Tensor):
        self.lib = _lib.load()
        self.B = shape[0]
        self.n = int(shape[0] * shape[1] * shape[2])
        f64 = dict(device=device, dtype=torch.float64)
        self.x_cur = torch.empty(shape, **f64)
        self.x_hat = torch.empty(shape, **f64)
        self.x_next = torch.empty(shape, **f64)
        self.d_cur = torch.empty(shape, **f64)
        self.x_in = torch.empty(shape, device=device, dtype=torch.float32)
        self.den = torch.empty(shape, device=device, dtype=torch.float32)
        self.sigma = torch.empty(self.B, device=device, dtype=torch.float32)
        self.sched = sched_host.to(device).contiguous()
        self.step = torch.zeros(1, device=device, dtype=torch.int32)

    def _s(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def init_from_latents(self, latents: Tensor, t0: float):
        _lib.check(self.lib.gecco_sampler_scale_f64(_vp(latents), t0, _vp(self.x_cur), self.n, self._s()), "sampler_scale")

    def churn(self, noise: Tensor, step_stride: int):
        _lib.check(self.lib.gecco_sampler_add_noise_f64(_vp(self.x_cur), _vp(noise), step_stride, _vp(self.sched),
                                                        _vp(self.step), 3, 1, _vp(self.x_hat), _vp(self.x_in),
                                                        _vp(self.sigma), self.n, self.B, self._s()), "sampler_churn")

    def redo(self, noise: Tensor):
        _lib.check(self.lib.gecco_sampler_add_noise_f64(_vp(self.x_cur), _vp(noise), 0, _vp(self.sched), _vp(self.step),
                                                        4, 0, _vp(self.x_cur), None, None, self.n, self.B, self._s()),
                   "sampler_redo")

    def euler(self):
        _lib.check(self.lib.gecco_sampler_euler_f64(_vp(self.x_hat), _vp(self.den), _vp(self.sched), _vp(self.step),
                                                    _vp(self.d_cur), _vp(self.x_next), _vp(self.x_in), _vp(self.sigma),
                                                    self.n, self.B, self._s()), "sampler_euler")

    def heun(self):
        _lib.check(self.lib.gecco_sampler_heun_f64(_vp(self.x_hat), _vp(self.x_next), _vp(self.den), _vp(self.d_cur),
                                                   _vp(self.sched), _vp(self.step), _vp(self.x_cur), self.n, self._s()),
                   "sampler_heun")

    def advance(self):
        _lib.check(self.lib.gecco_sampler_advance(_vp(self.step), 1, self._s()), "sampler_advance")


class Diffusion(_Base):
    """Backbone + conditioner + loss + reparameterisation (reference diffusion.py:168-470)."""

    def __init__(self, backbone: nn.Module, conditioner: Conditioner, loss: EDMLoss, reparam: Reparam = NoReparam(dim=3)):
        super().__init__()
        self.backbone = backbone
        self.conditioner = conditioner
        self.loss = loss
        self.reparam = reparam
        self.sampler_kwargs = dict(num_steps=64, sigma_min=0.002, sigma_max=self.sigma_max, rho=7, S_churn=0.5,
                                   S_min=0, S_max=float("inf"), S_noise=1, with_pbar=False)

    def extra_repr(self) -> str:
        return str(self.sampler_kwargs)

    @property
    def sigma_max(self) -> float:
        return self.loss.schedule.sigma_max

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=1e-4)

    def training_step(self, batch: Example, batch_idx):
        x, ctx = batch
        if torch.is_grad_enabled() and x.is_cuda:
            from .autograd import WEIGHT_IMAGES
            WEIGHT_IMAGES.prepare()   # the step's split-bf16 weight images in batched launches (recorded by the first step)
        loss = self.loss(self, x, ctx)
        self.log("train_loss", loss)
        return loss

    def validation_step(self, batch: Example, batch_idx):
        x, ctx = batch
        with torch.no_grad():
            loss = self.loss(self, x, ctx)
        self.log("val_loss", loss)

    @torch.no_grad()
    def graphed_forward(self, data: Tensor, sigma: Tensor, raw_context: Any | None = None, post_context: Any | None = None,
                        frozen_weights: bool = False):
        """One evaluation captured as a hipGraph (an addition to the reference API, for serving loops that evaluate the
        same shapes repeatedly: ~100 kernel launches replayed with one host call).  Returns `run(data=None, sigma=None)`:
        it copies new inputs into the captured buffers (when given), replays, and returns the captured output tensor.
        frozen_weights=True: the weight images are built by the warm-up call and the captured graph does not rebuild them —
        the graph serves the weight VALUES it was captured with (capture again after an update); default: every replay
        rebuilds them, so in-place weight updates are seen."""
        import contextlib
        x_buf, s_buf = data.clone(), sigma.clone()
        out = torch.empty_like(x_buf)
        if post_context is None and raw_context is not None:
            post_context = self.conditioner(raw_context)
        with (hip_ops.frozen_weights() if frozen_weights else contextlib.nullcontext()):
            self.forward(x_buf, s_buf, raw_context, post_context, out=out)   # warm-up: plans / workspaces outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.forward(x_buf, s_buf, raw_context, post_context, out=out)

        def run(data: Tensor | None = None, sigma: Tensor | None = None) -> Tensor:
            if data is not None:
                x_buf.copy_(data)
            if sigma is not None:
                s_buf.copy_(sigma)
            g.replay()
            return out
        run.graph = g
        return run

    # `torch.compile(model)` (example_configs/shapenet_airplane_unconditional.py:81): every FLOP of this forward runs in
    # libgecco_hip.so through ctypes, which dynamo cannot trace — the evaluation is excluded from tracing and runs as it
    # does in eager mode (same bits), instead of being cut into hundreds of graph breaks
    @torch.compiler.disable
    def forward(self, data: Tensor, sigma: Tensor, raw_context: Any | None, post_context: Any | None = None,
                do_cache: bool = False, cache: Any | None = None, out: Tensor | None = None) -> Tensor:
        if torch.is_grad_enabled() and data.is_cuda:
            from .autograd import WEIGHT_IMAGES
            WEIGHT_IMAGES.begin_forward()   # batched weight images serve the one forward that follows prepare() (training_step)
        if post_context is None:
            post_context = self.conditioner(raw_context)
        return self.backbone(data, sigma, raw_context, post_context, do_cache, cache, out=out)

    # ---- this model's own arithmetic mode / path switches (additions to the reference API; the reference's modules carry no
    # process-wide state, diffusion.py:160-178 — neither do ours once these are set: two models of different precision in one
    # serving process, on any host threads, do not see each other)
    def set_precision(self, name: str | None) -> "Diffusion":
        """Arithmetic of THIS model's evaluations ("fp32" | "bf16x3" | "fp16" | "mixed" | "w2"; None: the process-wide default,
        GECCO_PRECISION / hip_ops.set_default_precision)."""
        from .models.set_transformer import SetTransformer
        for m in self.modules():
            if isinstance(m, SetTransformer):
                m.set_precision(name)
        return self

    def set_option(self, name: str, value: int) -> "Diffusion":
        """Pin a library path switch (include/gecco_hip.h gecco_set_option) for THIS model; negative: follow the default again."""
        from .models.set_transformer import SetTransformer
        for m in self.modules():
            if isinstance(m, SetTransformer):
                m.set_option(name, value)
        return self

    @property
    def example_param(self) -> Tensor:
        return next(self.parameters())

    def t_steps(self, num_steps: int, sigma_max: float, sigma_min: float, rho: float) -> Tensor:
        return karras_t_steps(num_steps, sigma_max, sigma_min, rho).to(self.example_param.device)

    # ------------------------------------------------------------------------------------------ sampling
    @torch.no_grad()
    @_frozen_weights
    def sample_stochastic(self, shape: Sequence[int], context: Context3d | None, rng: torch.Generator = None,
                          noise: Tensor | Sequence[Tensor] | None = None, use_graph: bool = True, **kwargs) -> Tensor:
        """EDM stochastic Heun sampler (the paper's SDE sampler): num_steps steps = 2*num_steps - 1 evaluations.

        `noise` (optional): (num_steps + 1, *shape) tensor or sequence — entry 0 is the initial latent draw, entry
        i + 1 the churn noise of step i — replacing the generator draws (parity tests)."""
        kw = {**self.sampler_kwargs, **kwargs}
        num_steps = kw["num_steps"]
        device, dtype = self.example_param.device, self.example_param.dtype
        if dtype != torch.float32:
            raise NotImplementedError("the HIP denoiser computes in float32")
        shape = tuple(shape)
        if noise is None:
            if rng is None:
                rng = torch.Generator(device).manual_seed(42)
            # the reference's draw order on the same generator: the latents, then one draw per step (diffusion.py:300,324)
            # — so a seed / generator gives the clouds it gives there and is left in the same state — into one buffer
            noise = torch.empty((num_steps + 1, *shape), device=device, dtype=dtype)
            for i in range(num_steps + 1):
                torch.randn(shape, device=device, generator=rng, dtype=dtype, out=noise[i])
        elif not torch.is_tensor(noise):
            noise = torch.stack([n.to(device=device, dtype=dtype) for n in noise])
        noise = noise.to(device=device, dtype=dtype).contiguous()
        assert noise.shape == (num_steps + 1, *shape), noise.shape

        post_context = self.conditioner(context)
        ts = karras_t_steps(num_steps, kw["sigma_max"], kw["sigma_min"], kw["rho"])
        sched = build_schedule_table(ts, num_steps, kw["S_churn"], kw["S_min"], kw["S_max"], kw["S_noise"])
        st = _SamplerState(shape, device, sched)
        stride = st.n
        churn_noise = noise[1:]

        def evaluate():
            self(st.x_in, st.sigma, context, post_context, out=st.den)

        def full_step():      # steps 0 .. num_steps-2: Euler + 2nd-order correction
            st.churn(churn_noise, stride)
            evaluate()
            st.euler()
            evaluate()
            st.heun()
            st.advance()

        def last_step():      # t_next = 0: Euler only (reference diffusion.py:339)
            st.churn(churn_noise, stride)
            evaluate()
            st.euler()

        st.init_from_latents(noise[0], float(ts[0]))
        pbar = None
        if kw["with_pbar"]:
            from tqdm.auto import tqdm
            pbar = tqdm(total=num_steps, unit="step")
        if use_graph and num_steps > 2:
            st.x_in.zero_()
            st.sigma.fill_(1.0)
            evaluate()  # warm-up: allocates plans / workspaces outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                full_step()
            for _ in range(num_steps - 1):
                g.replay()
                if pbar:
                    pbar.update(1)
        else:
            for _ in range(num_steps - 1):
                full_step()
                if pbar:
                    pbar.update(1)
        last_step()
        if pbar:
            pbar.update(1)
            pbar.close()
        return self.reparam.diffusion_to_data(st.x_next, context)

    @torch.no_grad()
    def sample_ode(self, shape: Sequence[int], context: Context3d | None, rng: torch.Generator = None,
                   latents: Tensor | None = None, use_graph: bool = True, **kwargs) -> Tensor:
        """Deterministic (probability-flow ODE) Heun sampler: the EDM sampler without churn, what gecco-jax's
        `Diffusion.solve_sample_ode` integrates with diffrax's Heun solver on the schedule's time grid
        (gecco-jax models/diffusion.py:333-374); the torch package does not ship it (gecco-torch/README.md:49-52).  Same
        device loop as `sample_stochastic` with S_churn = 0: num_steps steps = 2 num_steps - 1 evaluations, fp64 state,
        one captured hipGraph per step.  `latents` (optional, (B, N, 3)): the starting noise instead of a generator draw.

        Where the two differ (parity unpinned: jax / diffrax are absent from the image).  diffrax's Heun applies the second-order
        correction on EVERY step of `StepTo(ts)`, and the JAX schedule's last time is its sigma_min > 0, where the trajectory ends
        (solve_sample_ode: t1 = ts[-1]); the EDM loop the torch reference ships (diffusion.py:271-352, what this method runs) appends
        t = 0 to the grid and takes that last step as a plain Euler step (the correction would divide by t_next = 0,
        diffusion.py:339).  Up to the last grid point the two integrate the same ODE with the same second-order scheme; this
        method's result is additionally pushed from sigma_min to 0 by one Euler step (= the denoiser's output at sigma_min)."""
        kw = {**self.sampler_kwargs, **kwargs, "S_churn": 0.0}
        num_steps = kw["num_steps"]
        device, dtype = self.example_param.device, self.example_param.dtype
        shape = tuple(shape)
        if latents is None:
            if rng is None:
                rng = torch.Generator(device).manual_seed(42)
            latents = torch.randn(shape, device=device, generator=rng, dtype=dtype)
        noise = torch.zeros((num_steps + 1, *shape), device=device, dtype=dtype)   # churn noise is multiplied by 0
        noise[0] = latents.to(device=device, dtype=dtype)
        return self.sample_stochastic(shape, context, noise=noise, use_graph=use_graph, **kw)

    @torch.no_grad()
    @_frozen_weights
    def sample_inpaint(self, known: Tensor, m_to_inpaint: int, context: Context3d | None = None, num_substeps: int = 1,
                       seed: int | None = 42, noise: Sequence[Tensor] | None = None, **kwargs) -> Tensor:
        """Completion of partial clouds (gecco-jax models/stochastic.py:101-231, `sample_inpaint`): `known` (B, n, 3) data-space
        points are kept — re-noised to the current level at every sub-step — while `m_to_inpaint` new points are sampled
        jointly with them; returns the (B, m, 3) new points in data space (fp64).  Per step i, sub-step j: refresh the
        known part at sigma_i, churn, Euler step to sigma_{i+1}, 2nd-order correction when i < steps - 1, and between
        sub-steps noise back up from sigma_{i+1} to sigma_i.  The JAX package draws per-step keys; here a generator
        (`seed`) or the injected `noise` list supplies the draws in call order: [initial (B, m + n, 3)], then per
        (i, j): known-part noise (B, n, 3), churn noise (B, m + n, 3) [, redo noise (B, m + n, 3) when j < num_substeps - 1]."""
        kw = {**self.sampler_kwargs, **kwargs}
        num_steps = kw["num_steps"]
        device, dtype = self.example_param.device, self.example_param.dtype
        if dtype != torch.float32:
            raise NotImplementedError("the HIP denoiser computes in float32")
        rng = torch.Generator(device=device)
        if seed is not None:
            rng = rng.manual_seed(seed)
        it = iter(noise) if noise is not None else None

        def randn(shape):
            if it is not None:
                t = next(it).to(device=device, dtype=dtype).contiguous()
                assert tuple(t.shape) == tuple(shape), (t.shape, shape)
                return t
            return torch.randn(tuple(shape), device=device, dtype=dtype, generator=rng)

        known = known.to(device=device, dtype=dtype).contiguous()
        B, n, _ = known.shape
        m = int(m_to_inpaint)
        known_diff = self.reparam.data_to_diffusion(known, context).contiguous()
        post_context = self.conditioner(context)
        ts = karras_t_steps(num_steps, kw["sigma_max"], kw["sigma_min"], kw["rho"])
        sched = build_schedule_table(ts, num_steps, kw["S_churn"], kw["S_min"], kw["S_max"], kw["S_noise"])
        st = _SamplerState((B, m + n, 3), device, sched)
        lib = st.lib

        def refresh():   # x_cur[:, m:] = known_diff + randn * sigma_cur
            nz = randn(known.shape)
            _lib.check(lib.gecco_sampler_refresh_known_f64(_vp(st.x_cur), _vp(known_diff), _vp(nz), _vp(st.sched), _vp(st.step),
                                                           0, m, n, B, st._s()), "sampler_refresh_known")

        # x_init = [0 | known_diff] + randn * sigma_0  (stochastic.py:189-197)
        init = randn((B, m + n, 3))
        base = torch.zeros(B, m + n, 3, device=device, dtype=dtype)
        base[:, m:] = known_diff
        st.x_cur.copy_(base.double() + (init * float(ts[0])).double())
        for i in range(num_steps):
            for j in range(num_substeps):
                refresh()
                st.churn(randn((B, m + n, 3)), 0)
                self(st.x_in, st.sigma, context, post_context, out=st.den)
                st.euler()
                if i < num_steps - 1:
                    self(st.x_in, st.sigma, context, post_context, out=st.den)
                    st.heun()
                else:
                    st.x_cur.copy_(st.x_next)
                if j < num_substeps - 1:
                    st.redo(randn((B, m + n, 3)))
            st.advance()
        return self.reparam.diffusion_to_data(st.x_cur, context)[:, :m]

    @_frozen_weights
    def evaluate_logp(self, data: Tensor, context: Context3d | None = None, n_trace_samples: int = 1, seed: int | None = 42,
                      probes: Tensor | None = None, return_details: bool = False, **kwargs):
        """Log-likelihood of clouds under the probability-flow ODE (gecco-jax models/diffusion.py:446-540, `evaluate_logp`; the torch
        package does not ship it): the data are carried from sigma_min to sigma_max by Heun's method on the schedule's time grid while
        the change of log-density, -div(dx/dt), is integrated beside them with Hutchinson's estimator (models/diffusion.py:175-192:
        eps^T J eps for Rademacher probes eps, the SAME probes at every evaluation, as the JAX code's constant `noise_key` makes them);
        logp = log N(latent; 0, sigma_max^2) + the integrated divergence + the reparametrisation's log |det|.

        With the EDM schedule (sigma(t) = t, scale 1) dx/dt = (x - D(x; t)) / t, so eps^T J eps = (n - eps^T J_D eps) / t with n = 3 N:
        one evaluation of the denoiser and one vector-Jacobian product through it (the HIP autograd Functions' input gradients,
        `test_gradient_with_respect_to_the_noisy_cloud`) per probe and ODE stage.  data (B, N, 3) in data space -> (B,) fp64.
        `probes` (optional, (n_trace_samples, B, N, 3) of +-1) replaces the generator draw (parity tests).  The state is fp64, the
        network input fp32, as in the samplers.  Parity unpinned (jax / diffrax absent): the test compares with `oracle/cpu_ref.py`'s
        restatement on torch autograd.  The reparametrisation's log |det| comes from `Reparam.ladj_data_to_diffusion` (closed forms:
        NoReparam 0, GaussianReparam -N sum_d log sigma_d, UVLReparam per point from the pinhole projection, atanh and log-range —
        checked against torch's Jacobian of the oracle's restatement in tests/test_modules_cpu.py)."""
        kw = {**self.sampler_kwargs, **kwargs}
        num_steps = kw["num_steps"]
        device = self.example_param.device
        if self.example_param.dtype != torch.float32:
            raise NotImplementedError("the HIP denoiser computes in float32")
        data = data.to(device=device, dtype=torch.float32).contiguous()
        B, N, dim = data.shape
        if not hasattr(self.reparam, "ladj_data_to_diffusion"):
            raise NotImplementedError(f"evaluate_logp: {type(self.reparam).__name__} has no ladj_data_to_diffusion")
        with torch.no_grad():
            ladj = self.reparam.ladj_data_to_diffusion(data, context).to(device=device, dtype=torch.float64)
        if probes is None:
            gen = torch.Generator(device=device)
            if seed is not None:
                gen.manual_seed(seed)
            probes = torch.randint(0, 2, (n_trace_samples, B, N, dim), device=device, generator=gen).float() * 2 - 1
        probes = probes.to(device=device, dtype=torch.float32)
        assert probes.shape[1:] == data.shape, (probes.shape, data.shape)
        with torch.no_grad():
            post_context = self.conditioner(context)
            x = self.reparam.data_to_diffusion(data, context).double()
        ts = karras_t_steps(num_steps, kw["sigma_max"], kw["sigma_min"], kw["rho"])[:num_steps].flip(0)   # sigma_min ... sigma_max
        frozen = [q for q in self.parameters() if q.requires_grad]
        for q in frozen:
            q.requires_grad_(False)       # only the input carries a gradient: the weight-gradient kernels are skipped

        def field(t: float, xs: Tensor):
            """(dx/dt, d logp/dt) at (t, xs): fp64 tensors (B, N, 3), (B,)."""
            with torch.enable_grad():
                xg = xs.float().requires_grad_(True)
                D = self(xg, torch.full((B,), t, device=device, dtype=torch.float32), context, post_context)
                eje = torch.zeros(B, dtype=torch.float64, device=device)
                for k in range(probes.shape[0]):
                    g, = torch.autograd.grad((D * probes[k]).sum(), xg, retain_graph=k + 1 < probes.shape[0])
                    eje += (g.double() * probes[k].double()).sum((1, 2))
            eje /= probes.shape[0]
            return (xs - D.detach().double()) / t, (float(N * dim) - eje) / t

        try:
            delta = torch.zeros(B, dtype=torch.float64, device=device)
            traj = [x.clone()] if return_details else None
            for i in range(num_steps - 1):
                t0, t1 = float(ts[i]), float(ts[i + 1])
                h = t1 - t0
                k1, d1 = field(t0, x)
                k2, d2 = field(t1, x + h * k1)
                x = x + 0.5 * h * (k1 + k2)
                delta = delta + 0.5 * h * (d1 + d2)
                if traj is not None:
                    traj.append(x.clone())
        finally:
            for q in frozen:
                q.requires_grad_(True)
        smax = float(kw["sigma_max"])
        prior = -0.5 * ((x / smax) ** 2).sum((1, 2)) - N * dim * math.log(smax * math.sqrt(2.0 * math.pi))
        logp = prior + delta + ladj
        if not return_details:
            return logp
        return {"logp": logp, "prior_logp": prior, "delta_jacobian": delta, "delta_reparam": ladj, "latent": x,
                "trajectory_diff": torch.stack(traj)}

    @torch.no_grad()
    @_frozen_weights
    def upsample(self, data: Tensor, new_latents: Tensor | None = None, n_new: int | None = None,
                 context: Context3d | None = None, seed: int | None = 42, num_substeps=5,
                 noise: Sequence[Tensor] | None = None, use_graph: bool = False, **kwargs):
        """Generates `n_new` extra points conditionally independent given the per-layer inducer states of the known
        cloud (reference diffusion.py:354-470).  `noise` (optional): the randn draws in the reference's call order
        ([new_latents,] then per outer step: data noise, per sub-step churn noise [, redo noise]).

        One outer step — re-noise the known cloud, one full evaluation that builds the inducer cache, `num_substeps` x
        (churn, cached evaluation, Euler, cached evaluation, Heun[, redo]) on the new points, advance — is captured as ONE
        hipGraph (`use_graph=True`) and replayed per step; the step's noise is drawn before each replay, in the reference's
        order, into the buffers the graph reads (BASELINE config C5: "hipGraph, 128 steps").  Default: eager.  An outer step is
        ~1100 kernel nodes of 10 - 40 us each and the step is device-bound either way (the host issues it in half the time);
        measured on MI355X at C5 (8 clouds, 16 384 new points): 44.5 ms per outer step replayed against 42.6 ms eager, with one
        or two streams inside the capture alike (profiles/r03i_upsample_graph_vs_eager.txt) — the replay pays ~1.5 us per node
        that in-stream launches do not.  `sample_stochastic` (2 evaluations per graph at B = 64) gains from its graph."""
        kw = {**self.sampler_kwargs, **kwargs}
        num_steps = kw["num_steps"]
        device, dtype = self.example_param.device, self.example_param.dtype
        if dtype != torch.float32:
            raise NotImplementedError("the HIP denoiser computes in float32")
        rng = torch.Generator(device=device)
        if seed is not None:
            rng = rng.manual_seed(seed)
        it = iter(noise) if noise is not None else None

        def randn(shape, out=None):
            if it is not None:
                t = next(it).to(device=device, dtype=dtype).contiguous()
                assert tuple(t.shape) == tuple(shape), (t.shape, shape)
                return t if out is None else out.copy_(t)
            if out is not None:
                return torch.randn(tuple(shape), device=device, dtype=dtype, generator=rng, out=out)
            return torch.randn(tuple(shape), device=device, dtype=dtype, generator=rng)

        if (new_latents is None) == (n_new is None):
            raise ValueError("Either new_latents or n_new must be specified, but not both.")
        if new_latents is None:
            new_latents = randn((data.shape[0], n_new, data.shape[2]))
        assert isinstance(new_latents, Tensor)
        new_latents = new_latents.to(device=device, dtype=dtype).contiguous()

        data = self.reparam.data_to_diffusion(data.to(device=device, dtype=dtype).contiguous(), context)
        post_context = self.conditioner(context)  # once, not per evaluation (reference quirk diffusion.py:415-421)
        ts = karras_t_steps(num_steps, kw["sigma_max"], kw["sigma_min"], kw["rho"])
        sched = build_schedule_table(ts, num_steps, kw["S_churn"], kw["S_min"], kw["S_max"], kw["S_noise"])
        st = _SamplerState(tuple(new_latents.shape), device, sched)
        lib = st.lib
        B = data.shape[0]
        data_ctx = torch.empty_like(data)
        sigma_d = torch.empty(B, device=device, dtype=torch.float32)
        st.init_from_latents(new_latents, float(ts[0]))
        # the draws of one outer step, in the buffers its kernels (and the captured graph) read
        nz_data = torch.empty_like(data)
        nz_churn = [torch.empty_like(new_latents) for _ in range(num_substeps)]
        nz_redo = [torch.empty_like(new_latents) for _ in range(max(num_substeps - 1, 0))]

        def draw(last: bool) -> None:   # the reference's call order: data noise, then per sub-step churn [, redo]
            randn(data.shape, out=nz_data)
            for u in range(num_substeps):
                randn(new_latents.shape, out=nz_churn[u])
                if u < num_substeps - 1 and not last:
                    randn(new_latents.shape, out=nz_redo[u])

        def outer(last: bool) -> None:
            _lib.check(lib.gecco_sampler_add_noise_f32(_vp(data), _vp(nz_data), 0, _vp(st.sched), _vp(st.step), 0, _vp(data_ctx),
                                                       _vp(sigma_d), data.numel(), B, st._s()), "sampler_add_noise_f32")
            _, cache = self(data_ctx, sigma_d, context, post_context, do_cache=True, cache=None)
            for u in range(num_substeps):
                st.churn(nz_churn[u], 0)
                self(st.x_in, st.sigma, context, post_context, cache=cache, out=st.den)
                st.euler()
                if not last:
                    self(st.x_in, st.sigma, context, post_context, cache=cache, out=st.den)
                    st.heun()
                else:
                    st.x_cur.copy_(st.x_next)
                if u < num_substeps - 1 and not last:
                    st.redo(nz_redo[u])
            st.advance()

        pbar = None
        if kw["with_pbar"]:
            from tqdm.auto import tqdm
            pbar = tqdm(total=num_steps, unit="step")
        graph = None
        if use_graph and num_steps > 2:
            # warm-up outside the capture: plans / workspaces of the full and of the cached evaluation (state untouched)
            data_ctx.copy_(data)
            sigma_d.fill_(1.0)
            st.x_in.zero_()
            st.sigma.fill_(1.0)
            _, wc = self(data_ctx, sigma_d, context, post_context, do_cache=True, cache=None)
            self(st.x_in, st.sigma, context, post_context, cache=wc, out=st.den)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outer(False)
        for i in range(num_steps):
            last = i == num_steps - 1
            draw(last)
            if graph is not None and not last:
                graph.replay()
            else:
                outer(last)
            if pbar:
                pbar.update(1)
        if pbar:
            pbar.close()
        return self.reparam.diffusion_to_data(st.x_cur, context)
