"""Plain-PyTorch composite of pointcloudlib_amd's dense layers -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The package runs its HIP kernels and nothing else.  Tests that want a second implementation of the SAME modules (same
parameters, same indices) -- fp64 copies of a module as the truth for one stack, CPU tensors in the `-m "not gpu"` suite, gloo
data-parallel checks -- import this file: it registers itself with ``pointcloudlib_amd.misc.layers`` under the backend name
``"torch"``, and a module whose ``.backend`` is set to ``"torch"`` then evaluates here:

* ``mlp_forward``       ``[Linear(bias) -> BatchNorm(train: batch mean, BIASED variance, eps 1e-5) -> (Leaky)ReLU] x L`` on the rows of x
                        (+ max over groups of ``group_max`` rows): the reference's ``nn.Conv(k=1)`` + ``nn.BatchNorm`` + ``nn.ReLU`` stacks
                        (networks/cls/pointnet2.py:18-31,:51-57) on channel-last rows;
* ``batch_norm_rows``   that BatchNorm alone, with Jittor's running-statistics rule ``r += (batch - r) * momentum`` on the biased
                        variance;
* ``sepconv_forward_x`` PointCNN's ``SepConv(X @ concat(F1, F2))`` (misc/layers.py:133-169, :504-505) as matmul + einsum.
"""
import torch
import torch.nn.functional as F

from pointcloudlib_amd.misc import layers as _layers


class TorchComposite:
    @staticmethod
    def batch_norm_rows(x2d, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5):
        if running_mean is not None:
            with torch.no_grad():
                var, mean = torch.var_mean(x2d, dim=0, unbiased=False)
                running_mean += (mean - running_mean) * momentum
                running_var += (var - running_var) * momentum
        return F.batch_norm(x2d, None, None, gamma, beta, True, 0.0, eps)

    @classmethod
    def mlp_forward(cls, mlp, x, group_max=None):
        lead = x.shape[:-1]
        y = x.reshape(-1, x.shape[-1])
        for i in range(mlp.n_layers):
            y = F.linear(y, mlp.weights[i], None if mlp.biases is None else mlp.biases[i])
            if mlp.bn:
                rm, rv = getattr(mlp, f"running_mean_{i}"), getattr(mlp, f"running_var_{i}")
                if mlp.training:
                    y = cls.batch_norm_rows(y, mlp.gammas[i], mlp.betas[i], rm, rv, mlp.momentum, mlp.eps)
                else:
                    y = F.batch_norm(y, rm, rv, mlp.gammas[i], mlp.betas[i], False, 0.0, mlp.eps)
            if i < mlp.n_layers - 1 or mlp.last_act:
                y = F.relu(y) if mlp.slope == 0.0 else F.leaky_relu(y, mlp.slope)
        y = y.reshape(*lead, y.shape[-1])
        if group_max is not None:
            assert x.shape[-2] == group_max
            y = y.max(dim=-2)[0]               # ``Var.argmax(dim)[1]``: the max VALUE (networks/cls/pointnet2.py:57)
        return y

    @staticmethod
    def sepconv_forward_x(mod, X, F1, F2):
        Fc = F1 if F2 is None else torch.cat((F1, F2), dim=-1)
        x = torch.matmul(X, Fc)                                                      # misc/layers.py:504-505
        B, P, K, C = x.shape
        y = torch.einsum("bpkc,cjk->bpcj", x, mod.depthwise).reshape(B, P, C * mod.dm) + mod.depthwise_bias    # depthwise (1,K) conv :151
        y = mod.pointwise(y)                                                          # 1x1 conv (+ ReLU) :152-158
        return mod.bn(y) if mod.bn is not None else y


_layers.register_reference_backend("torch", TorchComposite)


def use(model, backend="torch"):
    """Set ``backend`` on every module of ``model`` that has one; returns the model."""
    for m in model.modules():
        if hasattr(m, "backend"):
            m.backend = backend
    return model
