"""The RPN training step on the host-emulated kernels (CPU): gradients vs the oracle's autograd, the fused update,
and the world-size-2 data-parallel step over gloo."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hipemu"))
sys.path.insert(0, HERE)
import train_cases as T  # noqa: E402


@pytest.fixture(scope="module")
def rt():
    from emu_runtime import emu_runtime
    return emu_runtime()


def test_anchor_target_layer_mirror_matches_reference_rng(rt):
    """AnchorTargetLayer(...)(feat_h, feat_w, Variable(gt), Variable(img_info)) with the reference's own test geometry
    (tests/test_anchor_target_layer.py:18-28); the subsample consumes NumPy's global RNG exactly as the reference does."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.models import AnchorTargetLayer
    from oracle import frcnn_oracle as O
    gt = np.array([[[10, 10, 60, 200, 0], [50, 100, 210, 210, 1], [160, 40, 200, 70, 2]]], dtype=np.float32)
    info = np.array([[224, 224]], dtype=np.int32)
    layer = AnchorTargetLayer(16, [0.5, 1, 2], [8, 16, 32], runtime=rt)
    np.random.seed(7)
    labels, targets, inds, n_all = layer(14, 14, Variable(gt), Variable(info))
    np.random.seed(7)
    wl, wt, wi, wn = O.anchor_target_layer(14, 14, gt, info)
    assert n_all == wn == 14 * 14 * 9
    assert np.array_equal(rt.mem.to_numpy(inds), wi) and np.array_equal(rt.mem.to_numpy(labels), wl)
    assert np.allclose(rt.mem.to_numpy(targets), wt, rtol=2e-7, atol=1e-7)
    assert (wl == 1).sum() <= 128 and (wl >= 0).sum() <= 256          # the invariants the reference's test asserts (:76-77,88)


def test_bbox_overlaps_mirror(rt):
    from chainer_faster_rcnn_amd.models import bbox_overlaps
    from oracle import frcnn_oracle as O
    rs = np.random.RandomState(0)
    a = rs.uniform(0, 100, (40, 2)); b = rs.uniform(0, 100, (4, 2))
    boxes, q = np.hstack([a, a + 30]), np.hstack([b, b + 50])
    assert np.array_equal(rt.mem.to_numpy(bbox_overlaps(boxes, q, runtime=rt)), O.bbox_overlaps(boxes, q))
    with pytest.raises(ValueError):
        bbox_overlaps(boxes.astype(np.float32), q, runtime=rt)


def test_rpn_train_step_small(rt):
    T.check_small_step(rt)


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(HERE, "hipemu")); sys.path.insert(0, HERE)
        from emu_runtime import emu_runtime
        from chainer_faster_rcnn_amd.chainer_compat import Variable
        from chainer_faster_rcnn_amd.train import RPNTrainer, TorchComm
        import parity_cases as P
        rt = emu_runtime()
        params = T.small_params()
        info = np.array([[40, 56]], dtype=np.int32)

        def sample(i):
            rs = np.random.RandomState(100 + i)
            gt = P.gt_case(rs, 2, 40, 56)
            gt[0, :, 2] = np.minimum(gt[0, :, 0] + 20, 55); gt[0, :, 3] = np.minimum(gt[0, :, 1] + 20, 39)
            return rs.randn(1, 3, 40, 56).astype(np.float32), gt
        # data parallel: rank r takes image r (batch[i::n]), one all-reduce, identical update everywhere
        tr = RPNTrainer(T.build_small(rt, params), comm=TorchComm())
        x, gt = sample(rank)
        np.random.seed(5 + rank)
        tr.step(Variable(x), Variable(info), Variable(gt))
        w_dp = rt.mem.to_numpy(tr.W)
        # single-process reference: both images' gradients added, one update
        ref = RPNTrainer(T.build_small(rt, params))
        gsum = None
        for r in range(world):
            x, gt = sample(r)
            np.random.seed(5 + r)
            ref.forward_backward(Variable(x), Variable(info), Variable(gt))
            g = rt.mem.to_numpy(ref.G)
            gsum = g if gsum is None else gsum + g
        ref.G[...] = gsum
        ref.update()
        q.put((rank, bool(np.array_equal(w_dp, rt.mem.to_numpy(ref.W))), float(np.abs(w_dp).sum())))
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_gloo_world2():
    """N > 1 path on CPU: two ranks, gloo, gradients summed by ONE all_reduce, weights stay bit-identical across ranks
    and equal the single-process update with the summed gradient."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]


def _dp4_worker(rank, world, port, q):
    import hashlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(HERE, "hipemu")); sys.path.insert(0, HERE)
        from emu_runtime import emu_runtime
        from chainer_faster_rcnn_amd.chainer_compat import Variable
        from chainer_faster_rcnn_amd.train import RPNTrainer, TorchComm
        import parity_cases as P
        rt = emu_runtime()
        params = T.small_params()
        info = np.array([[40, 56]], dtype=np.int32)

        def sample(i):
            rs = np.random.RandomState(100 + i)
            gt = P.gt_case(rs, 2, 40, 56)
            gt[0, :, 2] = np.minimum(gt[0, :, 0] + 20, 55); gt[0, :, 3] = np.minimum(gt[0, :, 1] + 20, 39)
            return rs.randn(1, 3, 40, 56).astype(np.float32), gt
        comm = TorchComm()
        launched = []
        orig = comm.all_reduce_sum_async
        comm.all_reduce_sum_async = lambda buf: (launched.append(int(buf.shape[0])), orig(buf))[1]
        tr = RPNTrainer(T.build_small(rt, params), comm=comm)
        x, gt = sample(rank)
        np.random.seed(5 + rank)
        for _ in range(2):                                         # two steps: the second reuses the bucket plan and the works list starts empty again
            tr.step(Variable(x), Variable(info), Variable(gt))
        w_dp = rt.mem.to_numpy(tr.W)
        # float64 reference of the same two steps over the four images (the ring's fp32 summation order is gloo's, not ours: compare within rounding)
        ref = RPNTrainer(T.build_small(rt, params))
        for _ in range(2):
            gsum = None
            for r in range(world):
                xr, gr = sample(r)
                np.random.seed(5 + r)
                ref.forward_backward(Variable(xr), Variable(info), Variable(gr))
                g = rt.mem.to_numpy(ref.G).astype(np.float64)
                gsum = g if gsum is None else gsum + g
            ref.G[...] = gsum.astype(np.float32)
            ref.update()
        w_ref = rt.mem.to_numpy(ref.W)
        err = float(np.abs(w_dp - w_ref).max() / max(np.abs(w_ref).max(), 1e-12))
        sizes = [int(e - s0) for _, s0, e in tr.buckets]
        q.put((rank, hashlib.sha1(w_dp.tobytes()).hexdigest(), err, launched == sizes * 2, len(tr.buckets), [b[0] for b in tr.buckets]))
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_gloo_world4():
    """Four ranks (rank > 2 ordering, bucket keying): every rank launches the SAME three buckets in the same (backward) order
    in both steps -- a collective is matched by call order, so a rank-dependent order would deadlock or mix buckets --, the weights stay bit-identical
    on all four ranks after two steps, and equal the float64-summed single-process update within fp32 rounding of the ring's summation order."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp4_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert len({h for _, h, _, _, _, _ in res}) == 1, res           # bit-identical weights everywhere
    assert all(err < 1e-6 for _, _, err, _, _, _ in res), res
    assert all(ok for _, _, _, ok, _, _ in res), res                # bucket launch order = plan order, on every rank, in both steps
    assert all(n == 3 for _, _, _, _, n, _ in res) and len({tuple(names) for *_, names in res}) == 1, res


def test_snapshot_roundtrip_after_training(rt, tmp_path):
    """save_npz / load_npz in chainer's link-path key scheme (forward.py:29): after one training step the packed weights are
    synced back, written, and a fresh model loaded from the file reproduces the trained model's RPN outputs bit for bit."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.serializers import load_npz, save_npz
    from chainer_faster_rcnn_amd.train import RPNTrainer
    import parity_cases as P
    rs = np.random.RandomState(0)
    params = T.small_params()
    model = T.build_small(rt, params)
    tr = RPNTrainer(model)
    x = rs.randn(1, 3, 40, 56).astype(np.float32)
    gt = P.gt_case(rs, 2, 40, 56)
    gt[0, :, 2] = np.minimum(gt[0, :, 0] + 20, 55); gt[0, :, 3] = np.minimum(gt[0, :, 1] + 20, 39)
    info = np.array([[40, 56]], dtype=np.int32)
    np.random.seed(0)
    tr.step(Variable(x), Variable(info), Variable(gt))
    path = str(tmp_path / "rpn_model_snapshot_1.npz")
    save_npz(path, model, trainer=tr)
    with np.load(path) as f:
        keys = set(f.files)
        assert {"trunk/conv1_1/W", "trunk/conv2_2/b", "RPN/rpn_conv_3x3/W", "RPN/rpn_cls_score/W", "RPN/rpn_bbox_pred/b"} <= keys
        assert f["trunk/conv2_1/W"].shape == (64, 64, 3, 3) and f["RPN/rpn_cls_score/W"].shape == (18, 64, 1, 1)
        assert not np.array_equal(f["trunk/conv2_1/W"], params["trunk/conv2_1/W"])          # the step did change the weights
    fresh = load_npz(path, T.build_small(rt, params))
    model.rpn_train = False
    fresh.rpn_train = False
    a = model.RPN.heads(model.trunk(Variable(x)))
    b = fresh.RPN.heads(fresh.trunk(Variable(x)))
    for u, v in zip(a[1:], b[1:]):
        assert np.array_equal(rt.mem.to_numpy(u), rt.mem.to_numpy(v))


def test_proposal_target_layer_mirror(rt):
    """ProposalTargetLayer(...)(proposals, Variable(gt)) vs the golden vectors the reference itself produced (tests/make_golden.py)."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.models import ProposalTargetLayer
    g = np.load(os.path.join(HERE, "golden", "proposal_target.npz"))
    layer = ProposalTargetLayer(16, [0.5, 1, 2], [8, 16, 32], 21, runtime=rt)
    for tag in "abc":
        np.random.seed(int(g[tag + "_seed"]))
        ug, ext, keep = layer(g[tag + "_props"], Variable(g[tag + "_gt"]))
        assert np.array_equal(rt.mem.to_numpy(keep), g[tag + "_keep"])
        assert np.array_equal(rt.mem.to_numpy(ug), g[tag + "_use_gt"])
        assert np.allclose(rt.mem.to_numpy(ext), g[tag + "_ext"], rtol=1e-6, atol=1e-7)
        assert len(g[tag + "_keep"]) <= 128


def test_rcnn_train_step_small(rt):
    T.check_small_rcnn_step(rt)


def test_rcnn_train_step_small_split_products(rt):
    """RCNNTrainer(conv_math="split"): the stage-2 step with the trunk's forward / input-gradient / weight-gradient convolutions as
    six bf16 MFMA products of 3-way split operands -- the same bars as the fp32-MFMA step."""
    T.check_small_rcnn_step(rt, conv_math="split")


def test_device_dropout_kernel_and_step(rt):
    """frcnn_dropout_f32: mask values are exactly 0 or 1/(1-ratio), y = x * mask, the keep rate is 1 - ratio, the draw is a pure function of
    (seed, index) -- and a stage-2 step that draws its masks on the device equals, bit for bit, the step that is handed those masks."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.train import RCNNTrainer
    rs = np.random.RandomState(0)
    x = rs.randn(300, 4096).astype(np.float32)
    y, m = rt.dropout(rt.mem.from_numpy(x), 0.5, 1234)
    y, m = rt.mem.to_numpy(y), rt.mem.to_numpy(m)
    assert set(np.unique(m)) == {0.0, 2.0} and np.array_equal(y, x * m)
    assert abs((m > 0).mean() - 0.5) < 0.004                                                  # 1.2 M draws: sigma = 4.5e-4
    assert abs((m.reshape(300, 4096) > 0).mean(axis=0) - 0.5).max() < 0.15 and abs((m[:, ::2] > 0).mean() - (m[:, 1::2] > 0).mean()) < 0.004
    y2, m2 = rt.dropout(rt.mem.from_numpy(x), 0.5, 1234)
    assert np.array_equal(rt.mem.to_numpy(m2), m)
    _, m3 = rt.dropout(rt.mem.from_numpy(x), 0.5, 1235)
    assert abs((rt.mem.to_numpy(m3) == m).mean() - 0.5) < 0.004                               # another seed: independent
    _, m4 = rt.dropout(rt.mem.from_numpy(x), 0.25, 7)
    m4 = rt.mem.to_numpy(m4)
    assert set(np.unique(m4)) == {0.0, np.float32(1.0 / 0.75)} and abs((m4 > 0).mean() - 0.75) < 0.004
    xg, gt, info = _step_inputs()
    grads = []
    for rng in ("device", "given"):
        model, _ = _small_full_model(rt)
        model.rcnn_train = True
        tr = RCNNTrainer(model, dropout_rng="device" if rng == "device" else "numpy", dropout_seed=3)
        np.random.seed(11)                                                                    # ProposalTargetLayer's subsample
        out = tr.forward_backward(Variable(xg), Variable(info), Variable(gt), masks=None if rng == "device" else masks)
        masks = tuple(rt.mem.to_numpy(v) for v in out["masks"])
        grads.append(rt.mem.to_numpy(tr.G).copy())
    assert np.array_equal(grads[0], grads[1]) and np.abs(grads[0]).max() > 0
    # the device stream is keyed on (seed, rank, iteration, forward within the iteration) -- a second forward of the same iteration draws fresh masks,
    # and a trainer REBUILT at iteration k (resume) draws what the original drew at iteration k
    model, _ = _small_full_model(rt)
    model.rcnn_train = True
    tr = RCNNTrainer(model, dropout_rng="device", dropout_seed=3)
    draws = []
    for it in (0, 0, 1):
        tr.iteration = it
        np.random.seed(11)
        draws.append(rt.mem.to_numpy(tr.forward_backward(Variable(xg), Variable(info), Variable(gt))["masks"][0]))
    assert not np.array_equal(draws[0], draws[1]) and not np.array_equal(draws[0], draws[2])
    model2, _ = _small_full_model(rt)
    model2.rcnn_train = True
    tr2 = RCNNTrainer(model2, dropout_rng="device", dropout_seed=3)
    tr2.iteration = 1
    np.random.seed(11)
    assert np.array_equal(rt.mem.to_numpy(tr2.forward_backward(Variable(xg), Variable(info), Variable(gt))["masks"][0]), draws[2])


def test_gradient_buckets_tile_the_flat_buffer(rt):
    """Data-parallel buckets: contiguous tail ranges of the flat gradient buffer in backward order, together covering it exactly
    once, each closed by a layer whose gradients are the last of the bucket to be produced."""
    from chainer_faster_rcnn_amd.train import RPNTrainer
    tr = RPNTrainer(T.build_small(rt, T.small_params()))
    assert 1 <= len(tr.buckets) <= 3
    end = tr.n_flat
    order = [n for n, _ in tr.convs]
    last_idx = len(order)
    for name, start, stop in tr.buckets:                      # backward order
        assert stop == end and start < stop and start == tr.seg[name + "/W"].offset
        assert order.index(name) < last_idx                   # closing layers come earlier and earlier in the network
        last_idx, end = order.index(name), start
    assert end == 0 and tr.buckets[-1][0] == order[0]


def test_rcnn_gradient_buckets_put_the_head_first(rt):
    """Stage 2: the head's gradients (fc6 alone is most of the buffer) are complete before the trunk's backward starts, so the
    first bucket is closed by a head layer and the buckets still tile the buffer."""
    from chainer_faster_rcnn_amd.train import RCNNTrainer
    params = T.small_params()
    params.update(T.small_head_params(np.random.RandomState(0)))
    model = T.build_small(rt, params)
    for n in RCNNTrainer.HEAD:
        getattr(model, n).set(params[n + "/W"], params[n + "/b"])
    tr = RCNNTrainer(model)
    assert tr.buckets[0][0] in RCNNTrainer.HEAD and tr.buckets[0][2] == tr.n_flat
    end = tr.n_flat
    for _, start, stop in tr.buckets:
        assert stop == end and start < stop
        end = start
    assert end == 0



# ---- derived state and parameter arenas
def _small_full_model(rt, conv_dtype="f32", head_dtype="f32", seed=0):
    import functools
    from chainer_faster_rcnn_amd.models import FasterRCNN, VGG16Prev
    params = T.small_params()
    params.update(T.small_head_params(np.random.RandomState(seed)))
    model = FasterRCNN(trunk_class=functools.partial(VGG16Prev, layers=T.SMALL_LAYERS), rpn_in_ch=64, rpn_mid_ch=64, feat_stride=4,
                       anchor_scales=(2, 4, 8), runtime=rt, conv_dtype=conv_dtype, head_dtype=head_dtype)
    model.load_params(params)
    model.RPN.proposal_layer.RPN_MIN_SIZE = 4
    model.RPN.proposal_layer._min_size = 4
    return model, params


def _step_inputs(seed=0, h=40, w=56):       # (28 x 40 holds no anchor of scale >= 2 entirely: the reference raises there, and since round 6 so does the mirror)
    import parity_cases as P
    rs = np.random.RandomState(seed)
    x = rs.randn(1, 3, h, w).astype(np.float32)
    gt = P.gt_case(rs, 2, h, w)
    gt[0, :, 2] = np.minimum(gt[0, :, 0] + 20, w - 1); gt[0, :, 3] = np.minimum(gt[0, :, 1] + 20, h - 1)
    return x, gt, np.array([[h, w]], dtype=np.int32)


def test_load_npz_rebuilds_the_stacked_inference_head(rt, tmp_path):
    """load_npz() on a model that already ran inference: the stacked cls_score|bbox_pred GEMM must carry the LOADED rows
    (it kept the old ones: detections silently wrong)."""
    from chainer_faster_rcnn_amd.serializers import load_npz, save_npz
    a, pa = _small_full_model(rt, seed=0)
    b, pb = _small_full_model(rt, seed=7)                     # different head weights
    x, _, _ = _step_inputs()
    xd = rt.mem.from_numpy(x)
    want = {k: rt.mem.to_numpy(v) for k, v in b.forward_device(xd, 40, 56).items()}
    a.forward_device(xd, 40, 56)                              # a's stacked head exists now
    path = str(tmp_path / "b.npz")
    save_npz(path, b)
    load_npz(path, a)
    got = {k: rt.mem.to_numpy(v) for k, v in a.forward_device(xd, 40, 56).items()}
    assert not np.array_equal(pa["cls_score/W"], pb["cls_score/W"])
    for k in want:
        assert np.array_equal(want[k], got[k]), k


def test_two_trainers_on_one_model_keep_training_it(rt):
    """rpn -> rcnn -> rpn on ONE model (the reference's alternation): the second trainer re-points the links at its own buffer; the
    first must notice, re-adopt the current values and keep updating the LIVE weights (it trained an orphaned copy)."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.train import RCNNTrainer, RPNTrainer
    model, _ = _small_full_model(rt)
    x, gt, info = _step_inputs()
    model.rpn_train = True
    t1 = RPNTrainer(model)
    np.random.seed(0)
    t1.step(Variable(x), Variable(info), Variable(gt))
    model.rcnn_train = True
    t2 = RCNNTrainer(model)                                   # adopts trunk convs (and the head) into ITS buffer
    np.random.seed(1)
    t2.step(Variable(x), Variable(info), Variable(gt))
    link = model.trunk.links["conv2_1"]
    after_t2 = rt.mem.to_numpy(link.Wp).copy()
    assert rt.mem.within(link.Wp, t2.W) and not rt.mem.within(link.Wp, t1.W)
    model.rpn_train = True
    np.random.seed(2)
    t1.step(Variable(x), Variable(info), Variable(gt))
    assert rt.mem.within(link.Wp, t1.W)                       # re-adopted ...
    live = rt.mem.to_numpy(link.Wp)
    assert not np.array_equal(live, after_t2)                 # ... and the live weights moved
    # the step started from t2's result, not from t1's stale copy: one SGD step changes a weight by at most lr * (|g| + wd |w|) + momentum * |v|
    assert np.abs(live - after_t2).max() < 0.05
    # inference sees the trained weights: the link's arrays ARE the windows
    seg = t1.seg["conv2_1/W"]
    assert np.array_equal(live.ravel(), rt.mem.to_numpy(t1.W)[seg.offset:seg.offset + seg.size])


def test_a_new_trainer_supersedes_the_old_one_of_its_class(rt):
    """A trainer per stage / epoch must not pile up arenas on the model: a second RPNTrainer replaces the first in the model's registry (it adopted
    every parameter of the set, starting from the first one's trained values), an RCNNTrainer registers next to it, detach_trainer() syncs and forgets."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.train import RCNNTrainer, RPNTrainer
    model, _ = _small_full_model(rt)
    x, gt, info = _step_inputs()
    model.rpn_train = True
    t1 = RPNTrainer(model)
    np.random.seed(0)
    t1.step(Variable(x), Variable(info), Variable(gt))
    trained = rt.mem.to_numpy(model.trunk.links["conv2_1"].Wp).copy()
    t1b = RPNTrainer(model)                                           # e.g. the next epoch's trainer
    assert np.array_equal(rt.mem.to_numpy(model.trunk.links["conv2_1"].Wp), trained)          # adopted the trained values, not the initial ones
    np.random.seed(1)
    t1b.step(Variable(x), Variable(info), Variable(gt))
    assert model._trainers == [t1b]
    model.rcnn_train = True
    t2 = RCNNTrainer(model)
    np.random.seed(2)
    t2.step(Variable(x), Variable(info), Variable(gt))
    assert model._trainers == [t1b, t2] and model._last_trainer is t2
    live = rt.mem.to_numpy(model.trunk.links["conv2_1"].Wp).copy()
    model.detach_trainer(t2)
    assert model._trainers == [t1b] and model._last_trainer is t1b
    link = model.trunk.links["conv2_1"]
    assert np.array_equal(rt.mem.to_numpy(link.W).reshape(link.cout, -1), live.T)             # synced to Chainer's layout on the way out


def test_load_npz_writes_through_adopted_links(rt, tmp_path):
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.serializers import load_npz, save_npz
    from chainer_faster_rcnn_amd.train import RPNTrainer
    model, params = _small_full_model(rt)
    other, _ = _small_full_model(rt)
    other.trunk.links["conv2_1"].set(params["trunk/conv2_1/W"] * 2.0, params["trunk/conv2_1/b"])
    path = str(tmp_path / "o.npz")
    save_npz(path, other)
    model.rpn_train = True
    tr = RPNTrainer(model)
    load_npz(path, model)
    link = model.trunk.links["conv2_1"]
    assert rt.mem.within(link.Wp, tr.W)                       # still the trainer's window, now holding the loaded values
    assert np.array_equal(rt.mem.to_numpy(link.Wp), rt.mem.to_numpy(other.trunk.links["conv2_1"].Wp))


def test_bf16_copies_follow_the_optimizer(rt):
    """A bf16 model infers with the TRAINED weights after trainer.step(): Conv3x3.Wb, the stacked RPN heads and Linear.Wb are
    re-derived from the updated fp32 parameters (they stayed at their load-time values)."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.train import RPNTrainer
    model, params = _small_full_model(rt, conv_dtype="bf16", head_dtype="bf16")
    x, gt, info = _step_inputs()
    xd = rt.mem.from_numpy(x)
    before = rt.mem.to_numpy(model.forward_device(xd, 40, 56, keep=True)["feat"]).copy()
    model.rpn_train = True
    tr = RPNTrainer(model, lr=0.05)                           # a large step so bf16 rounding cannot hide the change
    np.random.seed(0)
    tr.step(Variable(x), Variable(info), Variable(gt))
    model.rpn_train = False
    after = rt.mem.to_numpy(model.forward_device(xd, 40, 56, keep=True)["feat"])
    assert not np.array_equal(before, after)
    # reference: a FRESH bf16 model loaded with the trained fp32 weights
    from chainer_faster_rcnn_amd.serializers import namedparams
    trained = {k: rt.mem.to_numpy(rt.mem.contiguous(v)) for k, v in namedparams(model)}
    fresh, _ = _small_full_model(rt, conv_dtype="bf16", head_dtype="bf16")
    fresh.load_params(trained)
    want = rt.mem.to_numpy(fresh.forward_device(xd, 40, 56, keep=True)["feat"])
    assert np.array_equal(after, want)


def test_alternation_then_bf16_inference_and_snapshot_see_both_trainers(rt, tmp_path):
    """rpn -> rcnn on ONE bf16 model: the RCNNTrainer is the last trainer but owns only the trunk and the FC head; the RPN's
    3x3 convolution and heads were trained by the RPNTrainer.  bf16 inference (derived copies) and a snapshot taken afterwards must carry
    BOTH trainers' weights: a fresh model loaded with the live fp32 parameters gives the same detections, and so does one loaded from
    the snapshot."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.serializers import load_npz, namedparams, save_npz, save_trainer_npz
    from chainer_faster_rcnn_amd.train import RCNNTrainer, RPNTrainer
    model, params = _small_full_model(rt, conv_dtype="bf16", head_dtype="bf16")
    x, gt, info = _step_inputs()
    xd = rt.mem.from_numpy(x)
    model.rpn_train = True
    t1 = RPNTrainer(model, lr=0.05)
    np.random.seed(0)
    t1.step(Variable(x), Variable(info), Variable(gt))
    model.rcnn_train = True
    t2 = RCNNTrainer(model, lr=0.05)
    np.random.seed(1)
    t2.step(Variable(x), Variable(info), Variable(gt))
    model.rcnn_train = False
    got = {k: rt.mem.to_numpy(v).copy() for k, v in model.forward_device(xd, 28, 40).items()}
    # the live fp32 parameters (read AFTER the inference above synced every trainer)
    live = {k: rt.mem.to_numpy(rt.mem.contiguous(v)).copy() for k, v in namedparams(model)}
    assert not np.array_equal(live["RPN/rpn_conv_3x3/W"], params["RPN/rpn_conv_3x3/W"])          # the RPN trainer's update is there
    assert not np.array_equal(live["fc6/W"], params["fc6/W"])                                      # and the RCNN trainer's
    fresh, _ = _small_full_model(rt, conv_dtype="bf16", head_dtype="bf16")
    fresh.load_params(live)
    want = {k: rt.mem.to_numpy(v) for k, v in fresh.forward_device(xd, 28, 40).items()}
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    path = str(tmp_path / "alt.npz")
    save_npz(path, model, trainer=t2)
    with np.load(path) as f:
        assert np.array_equal(f["RPN/rpn_conv_3x3/W"], live["RPN/rpn_conv_3x3/W"]) and np.array_equal(f["RPN/rpn_cls_score/W"], live["RPN/rpn_cls_score/W"])
    tpath = str(tmp_path / "alt_trainer.npz")
    save_trainer_npz(tpath, t2)
    with np.load(tpath) as f:
        assert np.array_equal(f["updater/model:main/RPN/rpn_conv_3x3/W"], live["RPN/rpn_conv_3x3/W"])
    other, _ = _small_full_model(rt, conv_dtype="bf16", head_dtype="bf16")
    load_npz(path, other)
    back = {k: rt.mem.to_numpy(v) for k, v in other.forward_device(xd, 28, 40).items()}
    for k in want:
        assert np.array_equal(back[k], want[k]), k


def test_bf16_convs_with_a_split_product_head(rt):
    """FasterRCNN(conv_dtype='bf16', head_dtype='f32s') (forward_device() raised on the channel-blocked map): pool5 comes
    from the blocked bf16 map in fp32 and is split for the head; same detections as the keep=True path, which pools the fp32 copy."""
    model, _ = _small_full_model(rt, conv_dtype="bf16", head_dtype="f32s")
    x, _, _ = _step_inputs()
    xd = rt.mem.from_numpy(x)
    a, b = model.forward_device(xd, 28, 40, keep=True), model.forward_device(xd, 28, 40)
    for k in ("n_out", "rois", "cls_prob", "pred_boxes"):
        assert np.array_equal(rt.mem.to_numpy(a[k]), rt.mem.to_numpy(b[k])), k


def test_bf16_inference_pools_from_the_blocked_map(rt):
    """forward_device() of a bf16 model (keep=False: the inference path) pools straight from the channel-blocked bf16 conv5_3 map;
    with keep=True the fp32 NCHW copy is made and pooled: the same detections, bit for bit."""
    model, _ = _small_full_model(rt, conv_dtype="bf16", head_dtype="bf16")
    x, _, _ = _step_inputs()
    xd = rt.mem.from_numpy(x)
    a = model.forward_device(xd, 40, 56, keep=True)
    b = model.forward_device(xd, 40, 56)
    assert getattr(model.trunk, "feat_shape", None) is not None
    for k in ("n_out", "rois", "cls_prob", "pred_boxes"):
        assert np.array_equal(rt.mem.to_numpy(a[k]), rt.mem.to_numpy(b[k])), k
    model2, _ = _small_full_model(rt, conv_dtype="bf16", head_dtype="f32")      # bf16 convs, fp32 head: fp32 pool5 from the blocked map
    a, b = model2.forward_device(xd, 40, 56, keep=True), model2.forward_device(xd, 40, 56)
    for k in ("n_out", "rois", "cls_prob", "pred_boxes"):
        assert np.array_equal(rt.mem.to_numpy(a[k]), rt.mem.to_numpy(b[k])), k


# ---- snapshots in chainer's on-disk scheme (fixtures hand-built by tests/make_chainer_snapshot.py)
def _fixture_model(rt):
    import functools
    from chainer_faster_rcnn_amd.models import FasterRCNN, VGG16Prev
    model = FasterRCNN(trunk_class=functools.partial(VGG16Prev, layers=T.SMALL_LAYERS), rpn_in_ch=64, rpn_mid_ch=64, feat_stride=4,
                       anchor_scales=(2, 4, 8), runtime=rt)
    model.RPN.proposal_layer.RPN_MIN_SIZE = 4
    model.RPN.proposal_layer._min_size = 4
    return model


def test_load_chainer_model_snapshot_fixture(rt, tmp_path):
    """forward.py:29 `serializers.load_npz('data/VGG16_faster_rcnn_final.model', model)`: a file in chainer's key / layout scheme
    (no .npz suffix, link-path keys, (out,in,kh,kw) / (out,in) arrays) drives inference; save_npz writes the same scheme back."""
    import shutil
    from chainer_faster_rcnn_amd.serializers import load_npz, namedparams, save_npz
    src = os.path.join(HERE, "golden", "chainer_model_snapshot_small.npz")
    path = str(tmp_path / "small_faster_rcnn_final.model")           # the reference's files carry no .npz suffix
    shutil.copy(src, path)
    with np.load(src) as f:
        want = {k: f[k] for k in f.files}
    model = load_npz(path, _fixture_model(rt))
    got = {k: rt.mem.to_numpy(rt.mem.contiguous(v)) for k, v in namedparams(model)}
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), k
    ref = _fixture_model(rt)
    ref.load_params(want)
    x, _, _ = _step_inputs()
    xd = rt.mem.from_numpy(x)
    a, b = model.forward_device(xd, 40, 56), ref.forward_device(xd, 40, 56)
    for k in a:
        assert np.array_equal(rt.mem.to_numpy(a[k]), rt.mem.to_numpy(b[k])), k
    out = str(tmp_path / "resaved.model")
    save_npz(out, model)
    assert os.path.exists(out) and not os.path.exists(out + ".npz")
    with np.load(out) as f:
        assert sorted(f.files) == sorted(want) and all(np.array_equal(f[k], want[k]) for k in want)


def test_trainer_snapshot_fixture_and_resume(rt, tmp_path):
    """train_rpn.py:101-105 `extensions.snapshot()`: parameters + MomentumSGD velocities + iteration in chainer's trainer tree; the
    iterator / extension entries of such a file are ignored.  A run resumed from a snapshot continues bit-identically."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.serializers import load_trainer_npz, save_trainer_npz
    from chainer_faster_rcnn_amd.train import RPNTrainer
    src = os.path.join(HERE, "golden", "chainer_trainer_snapshot_small.npz")
    with np.load(src) as f:
        want = {k: f[k] for k in f.files}
    model = _fixture_model(rt)
    with np.load(os.path.join(HERE, "golden", "chainer_model_snapshot_small.npz")) as f:
        model.load_params({k: f[k] for k in f.files})                # some other weights first
    model.rpn_train = True
    tr = load_trainer_npz(src, RPNTrainer(model))
    assert tr.iteration == 37
    w = tr.flat_to_chainer_layout(tr.W)
    v = tr.flat_to_chainer_layout(tr.V)
    for k in w:
        assert np.array_equal(w[k], want["updater/model:main/" + k]), k
        assert np.array_equal(v[k], want["updater/optimizer:main/" + k + "/v"]), k
    # resume == continue: step the loaded trainer, snapshot, load into a second trainer, step both once more
    x, gt, info = _step_inputs()
    np.random.seed(3)
    tr.step(Variable(x), Variable(info), Variable(gt))
    path = str(tmp_path / "rpn_trainer_snapshot_38")
    save_trainer_npz(path, tr)
    with np.load(path) as f:
        assert "updater/optimizer:main/trunk/conv2_1/W/v" in f.files and int(f["updater/iteration"]) == 38
        assert f["updater/optimizer:main/RPN/rpn_cls_score/W/v"].shape == (18, 64, 1, 1)
    model2 = _fixture_model(rt)
    with np.load(os.path.join(HERE, "golden", "chainer_model_snapshot_small.npz")) as f:
        model2.load_params({k: f[k] for k in f.files})
    model2.rpn_train = True
    tr2 = load_trainer_npz(path, RPNTrainer(model2))
    for t in (tr, tr2):
        np.random.seed(4)
        t.step(Variable(x), Variable(info), Variable(gt))
    assert tr2.iteration == tr.iteration == 39
    assert np.array_equal(rt.mem.to_numpy(tr.W), rt.mem.to_numpy(tr2.W)) and np.array_equal(rt.mem.to_numpy(tr.V), rt.mem.to_numpy(tr2.V))


def test_rpn_train_step_small_split_products(rt):
    """The same step with the forward and input-gradient convolutions on split tensors (csrc/conv_f32s.hip, training forms)."""
    T.check_small_step(rt, conv_math="split")


def test_trainers_across_image_sizes(rt):
    """A differently sized image every iteration (what train_rpn.py / train_rcnn.py feed): each step equals a new trainer's, bit for bit."""
    import train_cases as T
    assert T.check_trainers_across_image_sizes(rt) == 2


def test_an_image_without_proposals_gives_empty_outputs(rt):
    """An image on which no proposal survives ProposalLayer's min-size filter (proposal_layer.py:146-148: every clipped box is smaller than 16 px): the
    reference ends with zero RoIs, i.e. (0, 21) / (0, 84) outputs -- so does the mirror's `model(img, img_info)`, through RoI pooling and the head on no rows."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    rs = np.random.RandomState(0)
    params = T.small_params()
    params.update(T.small_head_params(rs))
    model = T.build_small(rt, params)
    for n in ("fc6", "fc7", "cls_score", "bbox_pred"):
        getattr(model, n).set(params[n + "/W"], params[n + "/b"])
    model.rpn_train = False
    x = rs.randn(1, 3, 8, 8).astype(np.float32)
    cls_prob, boxes = model(Variable(x), Variable(np.array([[8, 8]], np.int32)))
    assert tuple(cls_prob.shape) == (0, 21) and tuple(boxes.shape) == (0, 84)
    out = model.forward_device(rt.mem.from_numpy(x), 8, 8)
    assert int(rt.mem.to_numpy(out["n_out"])[0]) == 0 and not rt.mem.to_numpy(out["rois"]).any()
    # the oracle agrees that nothing survives
    from oracle import frcnn_oracle as O
    p, s = O.proposal_layer(np.zeros((1, 18, 2, 2), np.float32), np.zeros((1, 36, 2, 2), np.float32), np.array([[8, 8]], np.int32), feat_stride=4,
                            anchor_scales=(2, 4, 8))
    assert p.shape == (0, 4)


def test_proposal_target_layer_edge_behaviour(rt):
    """ProposalTargetLayer where the reference raises or returns nothing (checked against the live class when /root/reference is present: oracle/ref_harness):
    no ground-truth box -> NumPy's ValueError (argmax of an empty sequence, proposal_target_layer.py:91); no proposal -> the type check's AssertionError (:62); a
    ground truth that overlaps no proposal by 0.1 -> an EMPTY sample ((0, 5), (0, 84), (0,)) -- mirror, oracle and reference alike."""
    from chainer_faster_rcnn_amd.chainer_compat import Variable
    from chainer_faster_rcnn_amd.models.proposal_target_layer import ProposalTargetLayer
    from oracle import frcnn_oracle as O
    from oracle import ref_harness as rh
    rs = np.random.RandomState(0)
    props = np.abs(rs.randn(20, 4)).astype(np.float32) * 50
    props[:, 2:] += props[:, :2] + 20
    gt1 = np.array([[[10, 10, 60, 60, 3]]], np.float32)
    far = np.array([[[900, 900, 960, 960, 3]]], np.float32)
    ref = rh.load() if rh.available() else None

    def run(which, p, gt):
        np.random.seed(1)
        if which == "mirror":
            return [rt.mem.to_numpy(o) for o in ProposalTargetLayer(16, [0.5, 1, 2], [8, 16, 32], 21, runtime=rt)(p, Variable(gt))]
        if which == "oracle":
            return list(O.proposal_target_layer(p, gt))
        return list(ref.ProposalTargetLayer(16, [0.5, 1, 2], [8, 16, 32], 21)(p, ref.Variable(gt)))
    impls = ["mirror", "oracle"] + (["reference"] if ref is not None else [])
    for which in impls:
        with pytest.raises(ValueError, match="empty sequence"):
            run(which, props, np.zeros((1, 0, 5), np.float32))
        with pytest.raises(AssertionError):
            run(which, np.zeros((0, 4), np.float32), gt1)
        out = run(which, props, far)
        assert [tuple(o.shape) for o in out] == [(0, 5), (0, 84), (0,)], which
    base = run("oracle", props, gt1)
    for which in impls:
        out = run(which, props, gt1)
        assert all(np.array_equal(a, b) for a, b in zip(out, base)), which
