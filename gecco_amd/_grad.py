"""The HIP path is forward-only in this round (backward kernels are SURVEY.md 8(f) rank 1).  Rather than
silently returning tensors without a grad graph, every module refuses to run when autograd would
need one."""
import torch


class GeccoTrainingNotSupported(NotImplementedError):
    pass


def require_no_grad(module, *tensors) -> None:
    if not torch.is_grad_enabled():
        return
    needs = any(torch.is_tensor(t) and t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())
    if needs:
        raise GeccoTrainingNotSupported(
            f"{type(module).__name__}: the MI355X HIP path has no backward kernels yet; run under torch.no_grad() "
            "(sampling, upsampling, evaluation). Training support is the next scope row (SURVEY.md 8(f)).")
