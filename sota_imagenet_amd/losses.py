"""`CrossEntropyLoss` plugin: label-smoothed softmax cross entropy on float targets, one native HIP kernel.

Drop-in for `_target_: pytorch_tools.losses.smooth.CrossEntropyLoss` (sota_imagenet/arg_parser.py:140-142;
`smoothing: 0.1` in configs/hydra_exp/1.r50_baseline.yaml:34-35; built at train.py:81; call form
`criterion(output, target)` sota_imagenet/callbacks.py:316).  Targets are the loader's one-hot float rows
(dali_dataloader.py:123) or the soft rows CutMix/Mixup produce (callbacks.py:244-247); 1-D class indices are
accepted too.  Forward and the gradient wrt the logits come out of the same kernel launch.
"""
import torch
import torch.nn as nn

from . import ops


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, smoothing):
        loss, dlogits = ops.ce_loss(logits, target, smoothing, 1.0, need_grad=True)
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None, None


class CrossEntropyLoss(nn.Module):
    def __init__(self, mode="multiclass", smoothing=0.0, reduction="mean", from_logits=True, temperature=1.0, **unsupported):
        super().__init__()
        if mode != "multiclass" or reduction != "mean" or not from_logits or temperature != 1.0:
            raise NotImplementedError("only mode='multiclass', reduction='mean', from_logits=True, temperature=1 is on the hot path")
        for k, v in unsupported.items():
            if v not in (None, False, 0, 0.0):
                raise NotImplementedError(f"CrossEntropyLoss({k}={v!r}) unsupported")
        self.smoothing = float(smoothing)

    def forward(self, y_pred, y_true):
        if not y_pred.is_cuda:
            raise RuntimeError("CrossEntropyLoss: the MI355X hot path has no CPU fallback")
        y_pred = y_pred.float().contiguous()
        if y_true.dim() == 1:
            y_true = torch.nn.functional.one_hot(y_true.long(), y_pred.shape[1])
        y_true = y_true.to(torch.float32).contiguous()
        if torch.is_grad_enabled() and y_pred.requires_grad:
            return _CEFn.apply(y_pred, y_true, self.smoothing)
        loss, _ = ops.ce_loss(y_pred, y_true, self.smoothing, 1.0, need_grad=False)
        return loss
