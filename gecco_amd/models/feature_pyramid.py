"""Feature pyramid conditioner (API of reference models/feature_pyramid.py:17-73), ON THE DEVICE in channels-last.

The reference wraps torchvision's ConvNeXt-T / -S and returns NCHW maps; here `ConvNeXtExtractor` owns the same
parameters under the same state-dict keys (`stages.{s}.{0,1}...` = torchvision's `features[2s]`, `features[2s+1]`:
stem / downsample, then the CNBlocks with `block.{0,2,3,5}` and `layer_scale`) and runs them through the HIP path
(csrc/convnext.hip + the fused GEMM): every activation is (B, H, W, C), the returned maps are NCHW-shaped *views* of that
memory (torch.channels_last), which the projective lookup consumes without the NCHW -> NHWC transpose — and `upsample`,
which the reference makes re-run the conditioner per evaluation (diffusion.py:415-421), pays nothing for it.

torchvision is NOT needed (it is absent from the build image); `pretrained=True` fetches its weights when it is
installed, exactly like the reference, and raises otherwise — released GECCO checkpoints carry the conditioner's
weights in their state dict (`conditioner.stages...`), so `pretrained=False` + `load_state_dict` is the usual path."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Literal

import torch
from torch import Tensor, nn

from .. import _lib, hip_ops
from ..hip_ops import _ptr, _stream
from ..structs import Context3d

LN_EPS = 1e-6   # torchvision's ConvNeXt: partial(LayerNorm2d, eps=1e-6) / nn.LayerNorm(eps=1e-6)


@dataclass
class FeaturePyramidContext:
    features: list[Tensor]
    K: Tensor
    # channels-last copies for the HIP lookup, filled lazily by RayNetwork (once per conditioner call)
    _nhwc: list[Tensor] | None = field(default=None, repr=False, compare=False)


class FeaturePyramidExtractor(nn.Module):
    def forward(self, ctx_raw: Context3d) -> FeaturePyramidContext:
        raise NotImplementedError()


class _CNBlock(nn.Module):
    """Parameter container with torchvision's CNBlock names: block.0 dwconv, block.2 LayerNorm, block.3 / block.5 the
    pointwise linears, layer_scale (dim, 1, 1).  (Indices 1, 4, 6 are parameter-free there: Permute, GELU, Permute.)"""

    def __init__(self, dim: int, layer_scale: float):
        super().__init__()
        self.block = nn.Sequential(
            nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim, bias=True), nn.Identity(), nn.LayerNorm(dim, eps=LN_EPS),
            nn.Linear(dim, 4 * dim), nn.GELU(), nn.Linear(4 * dim, dim), nn.Identity())
        self.layer_scale = nn.Parameter(torch.ones(dim, 1, 1) * layer_scale)


# torchvision: convnext_tiny = [(96, 192, 3), (192, 384, 3), (384, 768, 9), (768, None, 3)], small: depths 3, 3, 27, 3
_SETTINGS = {"tiny": ((96, 3), (192, 3), (384, 9), (768, 3)), "small": ((96, 3), (192, 3), (384, 27), (768, 3))}


class ConvNeXtExtractor(FeaturePyramidExtractor):
    def __init__(self, n_stages: int = 3, model: Literal["tiny", "small"] = "tiny", pretrained: bool = True):
        super().__init__()
        if model not in _SETTINGS:
            raise ValueError(f"Unknown model {model}")
        if not 1 <= n_stages <= 3:
            raise ValueError("the HIP conditioner implements the first three stages (every shipped config uses n_stages=3)")
        self.stages = nn.ModuleList()
        prev = None
        for dim, depth in _SETTINGS[model][:n_stages]:
            if prev is None:   # stem: Conv2d(3, 96, k4, s4) + LayerNorm2d
                head = nn.Sequential(nn.Conv2d(3, dim, kernel_size=4, stride=4, bias=True), nn.LayerNorm(dim, eps=LN_EPS))
            else:              # downsample: LayerNorm2d + Conv2d(C, 2C, k2, s2)
                head = nn.Sequential(nn.LayerNorm(prev, eps=LN_EPS), nn.Conv2d(prev, dim, kernel_size=2, stride=2, bias=True))
            self.stages.append(nn.Sequential(head, nn.Sequential(*[_CNBlock(dim, 1e-6) for _ in range(depth)])))
            prev = dim
        for m in self.modules():   # torchvision's init: trunc_normal(0.02) weights, zero biases
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)
        if pretrained:
            try:
                import torchvision.models as tvm
            except ImportError as e:
                raise ImportError("pretrained=True fetches torchvision's ConvNeXt weights (as gecco_torch does); without "
                                  "torchvision construct with pretrained=False and load a GECCO checkpoint") from e
            tv = (tvm.convnext_tiny(weights=tvm.ConvNeXt_Tiny_Weights.DEFAULT) if model == "tiny"
                  else tvm.convnext_small(weights=tvm.ConvNeXt_Small_Weights.DEFAULT))
            sd = {}
            for k, v in tv.features.state_dict().items():
                i, rest = k.split(".", 1)
                if int(i) // 2 < n_stages:
                    sd[f"stages.{int(i) // 2}.{int(i) % 2}.{rest}"] = v
            self.load_state_dict(sd, strict=True)

    def forward(self, raw_ctx: Context3d) -> FeaturePyramidContext:
        """Channels-last forward on the HIP path.  With gradients enabled and trainable parameters it runs as autograd
        Functions (autograd.convnext_pyramid: the reference trains the conditioner with the denoiser); otherwise the fused
        inference sequence below."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from ..autograd import convnext_pyramid
            return FeaturePyramidContext(features=convnext_pyramid(self, raw_ctx.image), K=raw_ctx.K)
        with torch.no_grad():
            return self._forward_inference(raw_ctx)

    def _forward_inference(self, raw_ctx: Context3d) -> FeaturePyramidContext:
        lib = _lib.load()
        img = raw_ctx.image.float().contiguous()
        B, _, H, W = img.shape
        feats = []
        x = None
        for s, stage in enumerate(self.stages):
            head, blocks = stage[0], stage[1]
            if s == 0:
                conv, ln = head[0], head[1]
                C = conv.out_channels
                h, w = H // 4, W // 4
                x = torch.empty(B, h, w, C, device=img.device, dtype=torch.float32)
                _lib.check(lib.gecco_convnext_stem_f32(_ptr(img), _ptr(conv.weight), _ptr(conv.bias), _ptr(ln.weight), _ptr(ln.bias),
                                                       _ptr(x), B, H, W, C, LN_EPS, _stream()), "gecco_convnext_stem_f32")
            else:
                ln, conv = head[0], head[1]
                Cin, C = conv.in_channels, conv.out_channels
                h, w = x.shape[1] // 2, x.shape[2] // 2
                patches = torch.empty(B, h, w, 4 * Cin, device=img.device, dtype=torch.float32)
                _lib.check(lib.gecco_convnext_ln_patch2_f32(_ptr(x), _ptr(ln.weight), _ptr(ln.bias), _ptr(patches), B, x.shape[1],
                                                            x.shape[2], Cin, LN_EPS, _stream()), "gecco_convnext_ln_patch2_f32")
                # conv weight (2C, C, 2, 2) as the GEMM's (out, (dy, dx, c)) matrix (a small re-layout of parameters)
                wmat = conv.weight.permute(0, 2, 3, 1).reshape(C, 4 * Cin).contiguous()
                x = hip_ops.linear(patches.view(1, B * h * w, 4 * Cin), wmat, conv.bias, precision=self._precision()).view(B, h, w, C)
            rows = B * h * w
            for blk in blocks:
                dw, ln, pw1, pw2 = blk.block[0], blk.block[2], blk.block[3], blk.block[5]
                y = torch.empty_like(x)
                w_tap = dw.weight.reshape(C, 49).t().contiguous()   # tap-major (49, C): what the kernel stages in LDS
                _lib.check(lib.gecco_convnext_dwconv_ln_f32(_ptr(x), _ptr(w_tap), _ptr(dw.bias), _ptr(ln.weight), _ptr(ln.bias),
                                                            _ptr(y), B, h, w, C, LN_EPS, _stream()), "gecco_convnext_dwconv_ln_f32")
                hid = hip_ops.linear(y.view(1, rows, C), pw1.weight, pw1.bias, act="gelu", precision=self._precision())
                w2 = torch.empty_like(pw2.weight)
                b2 = torch.empty_like(pw2.bias)
                _lib.check(lib.gecco_convnext_fold_scale_f32(_ptr(pw2.weight), _ptr(pw2.bias), _ptr(blk.layer_scale.reshape(-1)),
                                                             _ptr(w2), _ptr(b2), C, 4 * C, _stream()), "gecco_convnext_fold_scale_f32")
                x = hip_ops.linear(hid, w2, b2, residual=x.view(1, rows, C), precision=self._precision()).view(B, h, w, C)
            feats.append(x.permute(0, 3, 1, 2))   # NCHW-shaped view of channels-last memory
        return FeaturePyramidContext(features=feats, K=raw_ctx.K)

    @staticmethod
    def _precision() -> str:
        p = hip_ops.default_precision()
        return "fp32" if p == "fp32" else "bf16x3"   # the pyramid feeds an fp32 gather: never the fp16 arithmetic
