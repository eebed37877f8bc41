"""fp16 helpers of the reference API (fp16_util.py).  ``use_fp16`` defaults to False in both
CLIs and the north star is fp32, so only ``zero_grad`` is on the hot path; the two
conversion helpers back ``UNetVideoModel.convert_to_fp16/32``.  The flat master-parameter helpers of the reference are not
provided: ``TrainLoop`` refuses ``use_fp16`` (out of scope)."""
import torch.nn as nn

_CONVS = (nn.Conv1d, nn.Conv2d, nn.Conv3d)


def convert_module_to_f16(module):
    if isinstance(module, _CONVS):
        module.weight.data = module.weight.data.half()
        module.bias.data = module.bias.data.half()


def convert_module_to_f32(module):
    if isinstance(module, _CONVS):
        module.weight.data = module.weight.data.float()
        module.bias.data = module.bias.data.float()


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
