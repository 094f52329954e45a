"""Autograd routing.  The fused inference entry points keep no intermediates; when autograd would record a call
(`needs_grad`) the modules switch to the unfused training path of gecco_amd/autograd.py (forward and backward both
in HIP).  A module without a backward must refuse (`require_no_grad`) instead of silently returning tensors without a
grad graph."""
import torch


class GeccoTrainingNotSupported(NotImplementedError):
    pass


def needs_grad(module, *tensors) -> bool:
    """True when autograd would record this call: grad mode on and a trainable parameter or input involved."""
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
