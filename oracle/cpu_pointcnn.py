"""CPU restatement of the reference's PointCNN blocks in the reference's own NCHW formulation -- TEST INFRASTRUCTURE
ONLY (PARITY UNPINNED: Jittor cannot be imported here; this follows the text of /root/reference/misc/layers.py).

Every function takes a ``pointcloudlib_amd.misc.pointcnn`` module (moved to the CPU) only as a *parameter container*
and recomputes its output the way misc/layers.py does: permute to ``[B,C,P,K]``, ``conv2d`` with ``(1,K)`` / 1x1
kernels, training-mode BatchNorm, permute back.  Neighbour indices and sampling come from ``pcl_oracle.c``.  What it
pins is the channel-last re-layout of the HIP-side modules (weight index mapping, concat order, activation/BN order).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as _o


def _bn(y, gamma, beta, eps=1e-5):
    """training-mode BatchNorm over every axis but 1 (NC.. layout), biased variance."""
    return F.batch_norm(y, None, None, gamma, beta, True, 0.0, eps)


def _act(y, on):
    return F.relu(y) if on else y


def dense_conv(mod, x_nc):
    """Dense_Conv1d / Dense_Conv2d (misc/layers.py:209-270) on ``[B,C,N]`` or ``[B,C,P,K]``: conv k=1 (bias) -> BN -> act."""
    m = mod.mlp
    assert m.n_layers == 1
    w, b = m.weights[0], m.biases[0]
    y = F.conv1d(x_nc, w[:, :, None], b) if x_nc.dim() == 3 else F.conv2d(x_nc, w[:, :, None, None], b)
    if m.bn:
        y = _bn(y, m.gammas[0], m.betas[0])
    return _act(y, m.last_act)       # dropout is the caller's business (tests run with p=0)


def conv_1xk(mod, x_nchw):
    """Conv (misc/layers.py:173-205): conv2d (1,K) (bias iff no BN) -> act -> BN."""
    lin = mod.linear
    K, cin = mod.K, mod.in_channels
    w = lin.weights[0].reshape(-1, K, cin).permute(0, 2, 1)[:, :, None, :]        # W[o,k*cin+d] -> w[o,d,0,k]
    b = None if lin.biases is None else lin.biases[0]
    y = _act(F.conv2d(x_nchw, w.contiguous(), b), lin.last_act)
    return _bn(y, mod.bn.weight, mod.bn.bias) if mod.bn is not None else y


def sep_conv(mod, x_nchw):
    """SepConv (misc/layers.py:133-169): depthwise (1,K) conv (bias) -> 1x1 conv (bias iff no BN) -> act -> BN."""
    C, dm, K = mod.depthwise.shape
    wd = mod.depthwise.reshape(C * dm, 1, 1, K)
    y = F.conv2d(x_nchw, wd, mod.depthwise_bias, groups=C)
    pw = mod.pointwise
    b = None if pw.biases is None else pw.biases[0]
    y = _act(F.conv2d(y, pw.weights[0][:, :, None, None], b), pw.last_act)
    return _bn(y, mod.bn.weight, mod.bn.bias) if mod.bn is not None else y


def xconv(mod, rep_pt, pts, fts):
    """XConv.execute (misc/layers.py:452-517): rep_pt [B,P,3], pts [B,P,K,3], fts [B,P,K,C]|None -> [B,P,C_out]."""
    B, P, K, _ = pts.shape
    pts_local = pts - rep_pt.unsqueeze(2).repeat(1, 1, K, 1)                       # :472-474
    pts_local = pts_local.permute(0, 3, 1, 2)                                       # :479
    d = mod.dense
    y = pts_local
    for i in range(2):                                                              # dense1, dense2 :480-483
        y = F.relu(_bn(F.conv2d(y, d.weights[i][:, :, None, None], d.biases[i]), d.gammas[i], d.betas[i]))
    fts_cat = y if fts is None else torch.cat((y, fts.permute(0, 3, 1, 2)), 1)      # :485-489
    x = conv_1xk(mod.x_trans_0, pts_local)                                          # [B,K*K,P,1]  :494
    x = dense_conv(mod.x_trans_1, x)
    X = dense_conv(mod.x_trans_2, x)
    X = X.permute(0, 2, 3, 1).reshape(B, P, K, K)                                   # :499-500
    fts_X = torch.matmul(X, fts_cat.permute(0, 2, 3, 1))                            # :504-505
    out = sep_conv(mod.end_conv, fts_X.permute(0, 3, 1, 2))                         # EndChannels :112-128
    return out.permute(0, 2, 3, 1).squeeze(2)                                       # :509


def pointcnn(mod, rep_pts, pts, fts, return_idx=False):
    """PointCNN.execute (misc/layers.py:388-411)."""
    if fts is not None and mod.dense is not None:
        fts = dense_conv(mod.dense, fts.permute(0, 2, 1)).permute(0, 2, 1)           # EndChannels1d(Dense_Conv1d) :393
    q = np.ascontiguousarray(rep_pts.detach().numpy().transpose(0, 2, 1))
    r = np.ascontiguousarray(pts.detach().numpy().transpose(0, 2, 1))
    idx = _o.knn(q, r, mod.K * mod.D)[:, 0::mod.D, :].transpose(0, 2, 1)             # :396-400
    li = torch.from_numpy(np.ascontiguousarray(idx).astype(np.int64))
    bi = torch.arange(pts.shape[0])[:, None, None]
    pts_regional = pts[bi, li]                                                       # select_region :379-386
    fts_regional = fts[bi, li] if fts is not None else None
    out = xconv(mod.x_conv, rep_pts, pts_regional, fts_regional)
    return (out, idx) if return_idx else out


def rand_pointcnn(mod, pts, fts, tie_stride):
    """RandPointCNN.execute (misc/layers.py:318-336)."""
    if 0 < mod.P < pts.shape[1]:
        _, rep = _o.fps(pts.detach().numpy(), mod.P, block_size=tie_stride, return_xyz=True)
        rep_pts = torch.from_numpy(rep).to(pts.dtype)          # (fp64 evaluation: the coordinates are exact fp32 values)
    else:
        rep_pts = pts
    return rep_pts, pointcnn(mod.pointcnn, rep_pts, pts, fts)


def rand_pointcnn_decoder(mod, x_l, x_h):
    """RandPointCNN_Decoder.execute (misc/layers.py:285-303)."""
    pts_l, fts_l = x_l
    pts_h, fts_h = x_h
    f = pointcnn(mod.pointcnn, pts_h, pts_l, fts_l)
    f = torch.cat((f, fts_h), dim=2)
    return pts_h, dense_conv(mod.conv_fuse, f.permute(0, 2, 1)).permute(0, 2, 1)


def pointcnn_cls(net, x, normal=None):
    """PointCNNcls.execute (networks/cls/pointcnn.py:39-53); dropout must be disabled (p=0) on ``net``."""
    ts = _o.optimal_block(x.shape[0])
    s = (x, x if normal is None else normal)
    s = rand_pointcnn(net.pcnn1, s[0], s[1], ts)
    for m in net.pcnn2:
        s = rand_pointcnn(m, s[0], s[1], ts)
    f = s[1].permute(0, 2, 1)
    for m in net.fcn:
        f = dense_conv(m, f)
    return f.mean(dim=2)


def pointcnn_partseg(net, x):
    """PointCNN_partseg.execute (networks/seg/pointcnn_partseg.py:33-47)."""
    ts = _o.optimal_block(x.shape[0])
    x_0 = rand_pointcnn(net.encoder_0, x, x, ts)
    x_1 = rand_pointcnn(net.encoder_1, *x_0, ts)
    x_2 = rand_pointcnn(net.encoder_2, *x_1, ts)
    x_3 = rand_pointcnn(net.encoder_3, *x_2, ts)
    x_3 = rand_pointcnn_decoder(net.decoder_0, x_3, x_3)
    x_2 = rand_pointcnn_decoder(net.decoder_1, x_3, x_2)
    x_1 = rand_pointcnn_decoder(net.decoder_2, x_2, x_1)
    x_0 = rand_pointcnn_decoder(net.decoder_3, x_1, x_0)
    return x_0[1].permute(0, 2, 1)
