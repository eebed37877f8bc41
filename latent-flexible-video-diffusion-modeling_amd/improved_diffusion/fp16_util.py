"""fp16 helpers of the reference API (fp16_util.py).  ``use_fp16`` defaults to False in both
CLIs and the north star is fp32, so only ``zero_grad`` is on the hot path; the conversion
helpers are kept for API compatibility."""
import torch.nn as nn
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors

_CONVS = (nn.Conv1d, nn.Conv2d, nn.Conv3d)


def convert_module_to_f16(module):
    if isinstance(module, _CONVS):
        module.weight.data = module.weight.data.half()
        module.bias.data = module.bias.data.half()


def convert_module_to_f32(module):
    if isinstance(module, _CONVS):
        module.weight.data = module.weight.data.float()
        module.bias.data = module.bias.data.float()


def make_master_params(model_params):
    flat = _flatten_dense_tensors([p.detach().float() for p in model_params])
    master = nn.Parameter(flat)
    master.requires_grad = True
    return [master]


def model_grads_to_master_grads(model_params, master_params):
    master_params[0].grad = _flatten_dense_tensors([p.grad.data.detach().float() for p in model_params])


def unflatten_master_params(model_params, master_params):
    return _unflatten_dense_tensors(master_params[0].detach(), list(model_params))


def master_params_to_model_params(model_params, master_params):
    model_params = list(model_params)
    for p, m in zip(model_params, unflatten_master_params(model_params, master_params)):
        p.detach().copy_(m)


def zero_grad(model_params):
    """Zero existing gradients in place (reference fp16_util.py:71-76).  One multi-tensor launch."""
    import torch as th
    grads = []
    for p in model_params:
        if p.grad is not None:
            p.grad.detach_()
            grads.append(p.grad)
    if grads:
        th._foreach_zero_(grads)
