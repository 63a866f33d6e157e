"""A second, independent witness for the oracle's [chainer-ext] operations (SURVEY.md header: Chainer v1 is not in /root/reference, so the
oracle's conv / pool / linear / softmax / loss restatements -- torch-CPU calls -- cannot be pinned to the reference itself).
Here every one of them is written out AGAIN as plain NumPy loops in float64, straight from the published definitions of the Chainer v1
functions the reference calls (file:line of the call site in each docstring) -- no torch, no shared code with oracle/ -- and the two
restatements have to agree to fp32 rounding on seeded inputs, including the cases where the definitions are easy to get wrong:
cross-correlation (not convolution), cover_all = ceil-mode pooling with ragged windows, the FIRST maximum of a window in the backward pass,
ignore_label = -1 normalised by the number of kept labels, Huber's two branches.  CPU only, seconds."""
import numpy as np

from oracle import frcnn_oracle as O


def conv2d_loops(x, W, b, pad):
    """L.Convolution2D(ci, co, k, stride 1, pad) (models/vgg16.py:39-68, region_proposal_network.py:53-57): y[n, o, i, j] = b[o] +
    sum_{c, ky, kx} W[o, c, ky, kx] * xpad[n, c, i + ky, j + kx] -- the kernel is NOT flipped."""
    n, c, h, w = x.shape
    co, _, k, _ = W.shape
    xp = np.zeros((n, c, h + 2 * pad, w + 2 * pad), np.float64)
    xp[:, :, pad:pad + h, pad:pad + w] = x
    oh, ow = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    y = np.zeros((n, co, oh, ow), np.float64)
    for ky in range(k):
        for kx in range(k):
            patch = xp[:, :, ky:ky + oh, kx:kx + ow]                          # (n, c, oh, ow)
            y += np.einsum("oc,nchw->nohw", W[:, :, ky, kx].astype(np.float64), patch)
    return y + b.astype(np.float64)[None, :, None, None]


def pool_windows(h, w):
    """F.MaxPooling2D(2, 2) = max_pooling_2d(ksize 2, stride 2, pad 0, cover_all=True) (models/vgg16.py:43): output size
    (h - k + s - 1) // s + 1 per axis -- every input cell is covered, the last window may hold one row / column."""
    oh, ow = (h - 2 + 2 - 1) // 2 + 1, (w - 2 + 2 - 1) // 2 + 1
    return oh, ow


def max_pool_loops(x):
    n, c, h, w = x.shape
    oh, ow = pool_windows(h, w)
    y = np.empty((n, c, oh, ow), np.float64)
    arg = np.empty((n, c, oh, ow, 2), np.int64)
    for i in range(oh):
        for j in range(ow):
            win = x[:, :, 2 * i:min(2 * i + 2, h), 2 * j:min(2 * j + 2, w)].astype(np.float64)
            flat = win.reshape(n, c, -1)
            a = flat.argmax(axis=2)                                         # NumPy's argmax: the FIRST maximum in row-major window order
            y[:, :, i, j] = np.take_along_axis(flat, a[..., None], axis=2)[..., 0]
            ww = win.shape[3]
            arg[:, :, i, j, 0], arg[:, :, i, j, 1] = 2 * i + a // ww, 2 * j + a % ww
    return y, arg


def test_convolution_is_cross_correlation_with_bias():
    rs = np.random.RandomState(0)
    for (ci, co, k, pad, h, w) in [(3, 8, 3, 1, 9, 11), (8, 5, 1, 0, 7, 6), (4, 6, 3, 1, 1, 5), (5, 4, 3, 1, 2, 2)]:
        x = rs.randn(1, ci, h, w).astype(np.float32)
        W = (rs.randn(co, ci, k, k) * 0.3).astype(np.float32)
        b = rs.randn(co).astype(np.float32)
        want = conv2d_loops(x, W, b, pad)
        got = O.conv2d(x, W, b, pad)
        assert got.shape == want.shape and got.dtype == np.float32
        assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1.0)
    # an asymmetric kernel on an impulse: the response is the kernel mirrored about the centre iff the operation is a correlation
    x = np.zeros((1, 1, 5, 5), np.float32)
    x[0, 0, 2, 2] = 1.0
    W = np.arange(9, dtype=np.float32).reshape(1, 1, 3, 3)
    y = O.conv2d(x, W, np.zeros(1, np.float32), 1)
    assert np.array_equal(y[0, 0, 1:4, 1:4], W[0, 0, ::-1, ::-1])


def test_max_pooling_covers_every_cell_and_its_gradient_goes_to_the_first_maximum():
    rs = np.random.RandomState(1)
    for (h, w) in [(6, 8), (7, 9), (1, 1), (5, 2), (75, 125 // 5)]:
        x = rs.randn(1, 3, h, w).astype(np.float32)
        want, arg = max_pool_loops(x)
        got = O.max_pool_2x2(x)
        assert got.shape == want.shape == (1, 3) + pool_windows(h, w)
        assert np.array_equal(got.astype(np.float64), want)
    # ties: a map of few distinct values -- the backward pass routes each window's gradient to its first maximum (window scan order)
    x = rs.randint(0, 3, size=(1, 2, 7, 9)).astype(np.float32)
    _, arg = max_pool_loops(x)
    dy = rs.randn(1, 2, 4, 5).astype(np.float32)
    want = np.zeros(x.shape, np.float64)
    for c in range(2):
        for i in range(4):
            for j in range(5):
                want[0, c, arg[0, c, i, j, 0], arg[0, c, i, j, 1]] += dy[0, c, i, j]
    assert np.array_equal(O.max_pool_2x2_backward(x, dy).astype(np.float64), want)


def test_convolution_gradients():
    """dx = dy correlated with the kernel rotated by 180 degrees and transposed, dW[o, c, ky, kx] = sum dy[o, i, j] xpad[c, i + ky, j + kx], db = sum dy."""
    rs = np.random.RandomState(2)
    ci, co, h, w, pad = 4, 5, 6, 7, 1
    x = rs.randn(1, ci, h, w).astype(np.float32)
    W = (rs.randn(co, ci, 3, 3) * 0.3).astype(np.float32)
    b = rs.randn(co).astype(np.float32)
    dy = rs.randn(1, co, h, w).astype(np.float32)
    xp = np.zeros((ci, h + 2, w + 2), np.float64)
    xp[:, 1:-1, 1:-1] = x[0]
    dW = np.zeros(W.shape, np.float64)
    dxp = np.zeros_like(xp)
    for ky in range(3):
        for kx in range(3):
            dW[:, :, ky, kx] = np.einsum("ohw,chw->oc", dy[0].astype(np.float64), xp[:, ky:ky + h, kx:kx + w])
            dxp[:, ky:ky + h, kx:kx + w] += np.einsum("oc,ohw->chw", W[:, :, ky, kx].astype(np.float64), dy[0].astype(np.float64))
    gx, gW, gb = O.conv2d_backward(x, W, b, dy, pad)
    assert np.abs(gx[0] - dxp[:, 1:-1, 1:-1]).max() <= 3e-6 * np.abs(dxp).max()
    assert np.abs(gW - dW).max() <= 3e-6 * np.abs(dW).max()
    assert np.abs(gb - dy[0].astype(np.float64).sum(axis=(1, 2))).max() <= 3e-6 * np.abs(dy).sum()


def test_linear_and_softmax():
    """L.Linear (models/faster_rcnn.py:33-36): y = x W^T + b on the row-major flattening of (R, C, 7, 7); F.softmax over axis 1 (faster_rcnn.py:178,
    region_proposal_network.py:117-120)."""
    rs = np.random.RandomState(3)
    x = rs.randn(5, 3, 2, 2).astype(np.float32)
    W = rs.randn(7, 12).astype(np.float32)
    b = rs.randn(7).astype(np.float32)
    want = np.array([[sum(float(x[r].ravel()[k]) * float(W[o, k]) for k in range(12)) + float(b[o]) for o in range(7)] for r in range(5)])
    got = O.linear(x.reshape(5, -1), W, b)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    s = rs.randn(4, 6, 3).astype(np.float32) * 5
    e = np.exp(s.astype(np.float64))
    assert np.abs(O.softmax(s, axis=1) - e / e.sum(axis=1, keepdims=True)).max() <= 2e-7


def test_rpn_losses_written_out():
    """models/region_proposal_network.py:160-204.  F.softmax_cross_entropy (Chainer v1: ignore_label -1, normalize=True): minus the mean, over the
    positions whose label is not -1, of log softmax at the label; F.huber_loss(delta): 0.5 d^2 where |d| < delta, delta (|d| - 0.5 delta) elsewhere,
    summed -- the reference divides the sum by the number of anchors K * A after its (4, A, K) -> (K, A, 4) re-interpretation of the channels."""
    rs = np.random.RandomState(4)
    fh, fw, A = 3, 4, 9
    n_all = fh * fw * A
    inds = np.sort(rs.choice(n_all, 40, replace=False))
    labels = rs.randint(-1, 2, size=40).astype(np.int32)
    score = rs.randn(1, 2 * A, fh, fw).astype(np.float32) * 2
    # anchor index k * A + a (k = y * fw + x) -> channels a (background) and A + a (foreground) at (y, x)
    total, kept, correct = 0.0, 0, 0
    for lab, idx in zip(labels, inds):
        if lab < 0:
            continue
        k, a = divmod(int(idx), A)
        y, x = divmod(k, fw)
        z = np.array([score[0, a, y, x], score[0, A + a, y, x]], np.float64)
        total -= z[lab] - np.log(np.exp(z).sum())
        kept += 1
        correct += int(int(z.argmax()) == lab)
    loss, acc = O.rpn_loss_cls(score, labels, inds, n_all, fh, fw, A)
    assert abs(float(loss) - total / kept) <= 1e-6 * max(total / kept, 1.0) and abs(float(acc) - correct / kept) <= 1e-7
    pred = (rs.randn(1, 4 * A, fh, fw) * 3).astype(np.float32)
    targets = rs.randn(40, 4).astype(np.float32)
    delta, tot = 3.0, 0.0
    flat = pred[0].reshape(4, A, fh * fw)                                      # channel = coord * A + a, as the reference re-interprets it
    for row, idx in enumerate(inds):
        k, a = divmod(int(idx), A)
        for coord in range(4):
            d = float(flat[coord, a, k]) - float(targets[row, coord])
            tot += 0.5 * d * d if abs(d) < delta else delta * (abs(d) - 0.5 * delta)
    got = float(O.rpn_loss_bbox(pred, targets, inds, A, delta))
    assert abs(got - tot / (fh * fw * A)) <= 2e-6 * tot / (fh * fw * A)


def test_momentum_sgd_with_weight_decay_written_out():
    """train_rpn.py:165-167: optimizer.add_hook(WeightDecay(0.0005)) runs before MomentumSGD(lr 0.001, momentum 0.9)'s update: g <- g + wd W;
    v <- momentum v - lr g; W <- W + v."""
    rs = np.random.RandomState(5)
    W, g, v = (rs.randn(6).astype(np.float32) for _ in range(3))
    W2, v2 = O.momentum_sgd_wd(W, g, v)
    for i in range(6):
        gi = np.float32(g[i] + np.float32(0.0005) * W[i])
        vi = np.float32(np.float32(0.9) * v[i] - np.float32(0.001) * gi)
        assert v2[i] == vi and W2[i] == np.float32(W[i] + vi)


def test_loss_gradients_written_out():
    """The analytic gradients of both loss pairs (region_proposal_network.py:160-204, faster_rcnn.py:152-164) against the oracle's autograd:
    d CE / d z = (softmax(z) - onehot) / kept; d Huber / d d = d where |d| < delta, delta sign(d) elsewhere, over the same denominators as the losses."""
    rs = np.random.RandomState(6)
    # stage 2: R sampled RoIs, 21 classes, 84 box outputs
    R, C = 9, 21
    z = (rs.randn(R, C) * 2).astype(np.float32)
    lab = rs.randint(0, C, size=R).astype(np.int32)
    bp = (rs.randn(R, 4 * C) * 1.5).astype(np.float32)
    tg = rs.randn(R, 4 * C).astype(np.float32)
    lc, lb, acc, dz, dbp = O.rcnn_loss_grads(z, bp, lab, tg, delta=1.0)
    z64 = z.astype(np.float64)
    sm = np.exp(z64 - z64.max(axis=1, keepdims=True))
    sm /= sm.sum(axis=1, keepdims=True)
    want_lc = -np.mean(np.log(sm[np.arange(R), lab]))
    d = bp.astype(np.float64) - tg
    want_lb = np.where(np.abs(d) < 1.0, 0.5 * d * d, np.abs(d) - 0.5).sum() / R
    onehot = np.zeros((R, C))
    onehot[np.arange(R), lab] = 1.0
    assert abs(float(lc) - want_lc) <= 2e-6 * want_lc and abs(float(lb) - want_lb) <= 2e-6 * want_lb
    assert abs(float(acc) - np.mean(z.argmax(axis=1) == lab)) <= 1e-7
    assert np.abs(dz - (sm - onehot) / R).max() <= 2e-7
    assert np.abs(dbp - np.clip(d, -1.0, 1.0) / R).max() <= 2e-7
    # RPN: gradients only where an inside anchor with a label >= 0 (CE) / any inside anchor (Huber) reads the map
    fh, fw, A = 3, 4, 9
    n_all = fh * fw * A
    inds = np.sort(rs.choice(n_all, 30, replace=False))
    labels = rs.randint(-1, 2, size=30).astype(np.int32)
    score = (rs.randn(1, 2 * A, fh, fw) * 2).astype(np.float32)
    pred = (rs.randn(1, 4 * A, fh, fw) * 3).astype(np.float32)
    targets = rs.randn(30, 4).astype(np.float32)
    _, _, ds, dp = O.rpn_loss_grads(score, pred, labels, targets, inds, n_all, fh, fw, A, delta=3.0, lam=1.0)
    want_ds, want_dp = np.zeros(score.shape), np.zeros(pred.shape)
    kept = int((labels >= 0).sum())
    for row, idx in enumerate(inds):
        k, a = divmod(int(idx), A)
        y, x = divmod(k, fw)
        if labels[row] >= 0:
            zz = np.array([score[0, a, y, x], score[0, A + a, y, x]], np.float64)
            p = np.exp(zz - zz.max())
            p /= p.sum()
            p[labels[row]] -= 1.0
            want_ds[0, a, y, x] += p[0] / kept
            want_ds[0, A + a, y, x] += p[1] / kept
        for coord in range(4):
            dd = float(pred[0, coord * A + a, y, x]) - float(targets[row, coord])
            want_dp[0, coord * A + a, y, x] += np.clip(dd, -3.0, 3.0) / (fh * fw * A)
    assert np.abs(ds - want_ds).max() <= 2e-7 and np.abs(dp - want_dp).max() <= 2e-7
