"""Training-step pieces of the reference's classification loop (train_cls.py) on PyTorch.

* ``soft_cross_entropy_loss`` -- train_cls.py:31-51 (label smoothing eps = 0.2); the reference builds the
  one-hot with a per-sample host sync (:41-42), here it is a device-side scatter.
* ``make_sgd`` -- ``nn.SGD(net.parameters(), lr, momentum)`` (train_cls.py:404): g += wd*p; v = mu*v + g;
  p -= lr*v, no dampening, no Nesterov == torch.optim.SGD.
"""
import torch
import torch.nn.functional as F


def soft_cross_entropy_loss(output, target, smoothing=True):
    target = target.reshape(-1).long()
    if not smoothing:
        return F.cross_entropy(output, target)
    eps = 0.2
    n_class = output.shape[1]
    one_hot = torch.zeros_like(output).scatter_(1, target[:, None], 1.0)
    one_hot = one_hot * (1 - eps) + (1 - one_hot) * eps / (n_class - 1)
    log_prb = F.log_softmax(output, dim=1)
    return -(one_hot * log_prb).sum(dim=1).mean()


def make_sgd(params, lr=0.02, momentum=0.9, weight_decay=0.0):
    params = list(params)
    # one multi-tensor kernel for the whole update on the GPU (same arithmetic as the default three-kernel foreach path)
    fused = bool(params) and all(p.is_cuda for p in params)
    return torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay, fused=fused)
