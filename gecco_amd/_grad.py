"""Autograd routing.  The fused inference entry points keep no intermediates; when autograd would record a call
(`needs_grad`) the modules switch to the unfused training path of gecco_amd/autograd.py (forward and backward both
in HIP).  A module without a backward must refuse (`require_no_grad`) instead of silently returning tensors without a
grad graph."""
import torch


class GeccoTrainingNotSupported(NotImplementedError):
    pass


def outside_fused_reach(module) -> bool:
    """True when the module tree holds a configuration the fused inference entry points do not take — today: a number of
    inducers other than 64 (the attention kernels are built around 64 = two 32-row matrix tiles; every shipped config uses
    64, the reference's constructor takes any), or a head dimension that is not a multiple of 8 up to 64 (multiples of 4 run
    here; anything else is rejected by the strided-batched GEMM's 16-byte loads).  Such models run the general composition of gecco_amd/autograd.py (every op
    still in libgecco_hip.so: attention with materialised scores on the strided-batched GEMM) instead of raising.
    Decided once per module object."""
    hit = module.__dict__.get("_gecco_general")
    if hit is None:
        def general(m):
            ind = getattr(m, "inducers", None)
            if ind is None or ind.dim() != 4:
                return False
            hd = ind.shape[3]                       # (1, H, I, head dim)
            return ind.shape[2] != 64 or hd % 8 != 0 or hd > 64
        hit = any(general(m) for m in module.modules())
        module.__dict__["_gecco_general"] = hit
    return hit


def needs_grad(module, *tensors) -> bool:
    """True when the call has to take the unfused path of gecco_amd/autograd.py: autograd would record it (grad mode on and a
    trainable parameter or input involved), or the configuration is outside the fused kernels' reach (`outside_fused_reach`:
    then also under no_grad — the Functions' forwards alone)."""
    if outside_fused_reach(module):
        return True
    if not torch.is_grad_enabled():
        return False
    return any(torch.is_tensor(t) and t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())


def require_no_grad(module, *tensors) -> None:
    if not torch.is_grad_enabled():
        return
    needs = any(torch.is_tensor(t) and t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())
    if needs:
        raise GeccoTrainingNotSupported(
            f"{type(module).__name__} has no backward on the MI355X HIP path (the denoiser, loss and optimizer do: "
            "gecco_amd/autograd.py, gecco_amd/optim.py); run this module under torch.no_grad() or keep its parameters frozen.")


def result_into(out, res, do_cache: bool):
    """The unfused path returns fresh tensors; a caller that passed `out=` (the samplers' state buffers) gets the denoised
    cloud written there, as the fused entry points do."""
    if out is None:
        return res
    den = res[0] if do_cache else res
    out.copy_(den)
    return (out, res[1]) if do_cache else out
