"""ORACLE (test infrastructure — never imported by the product path).

A SECOND, independent statement of the per-op arithmetic of oracle/ops_ref.py: plain fp64 numpy loops written from the
textbook definitions (no torch kernels, no autograd), used only by tests/test_oracle_naive.py to pin the torch-CPU oracle
at the fixture shapes.  The reference holds no vectors for this path ("parity unpinned"), so the two
restatements pin each other: a semantic slip in how ops_ref drives F.conv2d / F.batch_norm / F.max_pool2d (padding,
stride phase, biased vs unbiased variance, momentum convention, tie-breaking) would show up here.

What each function restates (reference call sites as in ops_ref.py):
  conv2d_*    cuDNN conv fwd / dgrad / wgrad under model(data), loss.backward()   sota_imagenet/callbacks.py:316-317
  bn_*        BatchNorm2d training forward / backward (+ residual, + ReLU)        train.py:76 (momentum), callbacks.py:316-317
  maxpool_*   MaxPool2d(3, 2, 1)                                                  callbacks.py:316-317
  smooth_ce   pytorch_tools.losses.smooth.CrossEntropyLoss on float targets       sota_imagenet/arg_parser.py:140-142
All tensors NHWC / KRSC numpy arrays; everything is computed in float64.
"""
import numpy as np


def _out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def conv2d_fwd(x, w, stride, pad):
    x, w = np.asarray(x, np.float64), np.asarray(w, np.float64)
    N, H, W, C = x.shape
    K, R, S, _ = w.shape
    Ho, Wo = _out(H, R, stride, pad), _out(W, S, stride, pad)
    y = np.zeros((N, Ho, Wo, K))
    for n in range(N):
        for oh in range(Ho):
            for ow in range(Wo):
                acc = np.zeros(K)
                for r in range(R):
                    ih = oh * stride - pad + r
                    if ih < 0 or ih >= H:
                        continue
                    for s in range(S):
                        iw = ow * stride - pad + s
                        if iw < 0 or iw >= W:
                            continue
                        acc += w[:, r, s, :] @ x[n, ih, iw, :]
                y[n, oh, ow] = acc
    return y


def conv2d_bwd(x, w, dy, stride, pad):
    """(dx, dw): scatter form — every product of the forward sum contributes its two partial derivatives."""
    x, w, dy = np.asarray(x, np.float64), np.asarray(w, np.float64), np.asarray(dy, np.float64)
    N, H, W, C = x.shape
    K, R, S, _ = w.shape
    Ho, Wo = dy.shape[1:3]
    dx, dw = np.zeros_like(x), np.zeros_like(w)
    for n in range(N):
        for oh in range(Ho):
            for ow in range(Wo):
                g = dy[n, oh, ow]  # [K]
                for r in range(R):
                    ih = oh * stride - pad + r
                    if ih < 0 or ih >= H:
                        continue
                    for s in range(S):
                        iw = ow * stride - pad + s
                        if iw < 0 or iw >= W:
                            continue
                        dx[n, ih, iw] += g @ w[:, r, s, :]
                        dw[:, r, s, :] += np.outer(g, x[n, ih, iw])
    return dx, dw


def bn_train(x, gamma, beta, running_mean, running_var, residual=None, relu=True, eps=1e-5, momentum=0.1):
    """(out, new_running_mean, new_running_var, batch mean, 1/sqrt(biased var + eps)); the running variance takes the
    UNBIASED batch variance, running = (1 - momentum) * running + momentum * batch."""
    x = np.asarray(x, np.float64)
    C = x.shape[-1]
    xf = x.reshape(-1, C)
    M = xf.shape[0]
    mean = xf.sum(0) / M
    var = ((xf - mean) ** 2).sum(0) / M
    invstd = 1.0 / np.sqrt(var + eps)
    y = (x - mean) * invstd * np.asarray(gamma, np.float64) + np.asarray(beta, np.float64)
    if residual is not None:
        y = y + np.asarray(residual, np.float64)
    if relu:
        y = np.where(y > 0, y, 0.0)
    rm = (1 - momentum) * np.asarray(running_mean, np.float64) + momentum * mean
    rv = (1 - momentum) * np.asarray(running_var, np.float64) + momentum * var * M / (M - 1)
    return y, rm, rv, mean, invstd


def bn_train_bwd(x, gamma, beta, dout, residual=None, relu=True, eps=1e-5):
    """(dx, dgamma, dbeta, dresidual) of out = relu(bn(x) + residual)."""
    x, dout = np.asarray(x, np.float64), np.asarray(dout, np.float64)
    gamma = np.asarray(gamma, np.float64)
    C = x.shape[-1]
    out, _, _, mean, invstd = bn_train(x, gamma, beta, np.zeros(C), np.ones(C), residual, relu, eps)
    dz = np.where(out > 0, dout, 0.0) if relu else dout
    M = x.size // C
    xhat = (x - mean) * invstd
    dbeta = dz.reshape(-1, C).sum(0)
    dgamma = (dz * xhat).reshape(-1, C).sum(0)
    dx = gamma * invstd * (dz - dbeta / M - xhat * dgamma / M)
    return dx, dgamma, dbeta, (dz if residual is not None else None)


def maxpool(x):
    """3x3, stride 2, pad 1; padding never wins (it is -inf)."""
    x = np.asarray(x, np.float64)
    N, H, W, C = x.shape
    Ho, Wo = _out(H, 3, 2, 1), _out(W, 3, 2, 1)
    y = np.full((N, Ho, Wo, C), -np.inf)
    arg = np.zeros((N, Ho, Wo, C, 2), np.int64)
    for oh in range(Ho):
        for ow in range(Wo):
            for r in range(3):
                ih = 2 * oh - 1 + r
                if ih < 0 or ih >= H:
                    continue
                for s in range(3):
                    iw = 2 * ow - 1 + s
                    if iw < 0 or iw >= W:
                        continue
                    v = x[:, ih, iw, :]
                    better = v > y[:, oh, ow, :]  # strict: the FIRST maximum in window scan order wins
                    y[:, oh, ow, :] = np.where(better, v, y[:, oh, ow, :])
                    arg[:, oh, ow, :, 0] = np.where(better, ih, arg[:, oh, ow, :, 0])
                    arg[:, oh, ow, :, 1] = np.where(better, iw, arg[:, oh, ow, :, 1])
    return y, arg


def maxpool_bwd(x, dy):
    x, dy = np.asarray(x, np.float64), np.asarray(dy, np.float64)
    _, arg = maxpool(x)
    dx = np.zeros_like(x)
    N, Ho, Wo, C = dy.shape
    for n in range(N):
        for oh in range(Ho):
            for ow in range(Wo):
                for c in range(C):
                    dx[n, arg[n, oh, ow, c, 0], arg[n, oh, ow, c, 1], c] += dy[n, oh, ow, c]
    return dx


def smooth_ce(logits, target, smoothing):
    """(loss, dloss/dlogits): mean over rows of (1-s) * -(sum_c y log p) + s * -(mean_c log p)."""
    z, t = np.asarray(logits, np.float64), np.asarray(target, np.float64)
    N, C = z.shape
    zs = z - z.max(1, keepdims=True)
    logp = zs - np.log(np.exp(zs).sum(1, keepdims=True))
    loss = ((1 - smoothing) * -(logp * t).sum(1) + smoothing * -logp.mean(1)).mean()
    p = np.exp(logp)
    # d/dz of -(sum_c a_c log p_c) = p * sum(a) - a, with a = (1-s) * t + s / C
    a = (1 - smoothing) * t + smoothing / C
    dz = (p * a.sum(1, keepdims=True) - a) / N
    return loss, dz


def sgd_steps(p, grads, lr, momentum, weight_decay):
    """torch.optim.SGD, dampening 0, no nesterov: g += wd * p; buf = g (first step) or mu * buf + g; p -= lr * buf."""
    p = np.asarray(p, np.float64).copy()
    buf = None
    for g in grads:
        g = np.asarray(g, np.float64) + weight_decay * p
        buf = g.copy() if buf is None else momentum * buf + g
        p = p - lr * buf
    return p, buf
