"""hipGraph replay of the inference forward: one captured graph instead of ~45 launches issued from Python.

The reference's per-image loop (forward.py:92-95) is launch-bound once the kernels are fast: a VGG-16 forward is ~45 HIP
launches of 5-450 us each, and a Python host needs 4-5 ms to issue them (measured: 191 img/s eager vs 258 img/s replayed on
the same MI355X).  `CapturedForward` records `model.forward_device` once for a fixed image shape into a hipGraph
(`torch.cuda.CUDAGraph` is the capture/replay plumbing; every node is one of this package's HIP kernels) and replays it per
image: the new image is copied into the captured input buffer, the outputs are the captured output buffers.

No fallback: without a GPU the capture raises like every other entry point of the package.
"""
import torch


class CapturedForward(object):
    def __init__(self, model, x, im_h, im_w, warmup=2):
        """x: (1, 3, H, W) float32 device array of the shape to capture for; im_h, im_w: the image size ProposalLayer clips to."""
        self.model, self.im_h, self.im_w = model, int(im_h), int(im_w)
        self.x = x.clone()                                  # the graph reads THIS buffer on every replay
        for _ in range(max(1, int(warmup))):                # workspaces are allocated and packed weights built outside the capture
            model.forward_device(self.x, self.im_h, self.im_w)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads (e.g. a process group's watchdog) may touch the runtime while this one captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = model.forward_device(self.x, self.im_h, self.im_w)
        # The graph holds RAW pointers into the runtime's workspaces (conv stream-K counter page and partials, proposal / NMS /
        # linear scratch).  Runtime.workspace() drops a workspace when a larger one is requested (a bigger image, a training step):
        # keep every array that existed at capture time alive for the lifetime of this object, so replays never write into
        # memory the allocator has handed to someone else.
        self._pinned_workspaces = dict(model.rt._ws)

    def replay(self, x=None):
        """Run the captured forward; `x` (same shape) replaces the input first.  Returns the dict of captured output arrays
        (cls_prob, pred_boxes, rois, probs, n_out) -- overwritten by the next replay."""
        if x is not None:
            self.x.copy_(x)
        self.graph.replay()
        return self.out


class ForwardsInFlight(object):
    """n (default 2) images in flight on one GPU: n model instances -- each with its own Runtime (scratch workspaces) and its own captured graph, all with the
    same weights -- replay on n HIP streams, image k on slot k % n.  The hardware interleaves image k's proposal / RoI-pooling / FC stages (launch-latency- and
    weight-streaming-bound: most of the chip idles) with image k + 1's convolutions: measured on the MI355X 1640 -> 2093 img/s for the bf16 network (x 1.28),
    275 -> 283-285 for fp32, 419 -> 448 for the split-product fp32 network; three in flight is slower than two (1911).  Each image still takes its full forward:
    this is the throughput form of a serving loop (the reference's forward.py:85-94 handles one image at a time; nothing in it forbids the next one from starting).

    The HIP runtime maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and two streams that share a queue run one after the other; which
    streams of a process share one depends on how many it has created before.  The constructor therefore takes `n_candidates` streams and keeps the set that
    overlaps best in a short probe (`probe_steps` replays per candidate set).

    make_model(runtime) -> a FasterRCNN bound to that runtime, parameters loaded.  submit(x) enqueues one image and returns (slot, outputs) -- device arrays that
    are valid once `wait(slot)` (or a device synchronise) has returned and until the slot's next submit."""

    def __init__(self, make_model, runtime_factory, x, im_h, im_w, n=2, n_candidates=6, probe_steps=40):
        import itertools
        import time
        self.n = int(n)
        self.slots = []
        for _ in range(self.n):
            rt = runtime_factory()
            model = make_model(rt)
            self.slots.append(CapturedForward(model, x, im_h, im_w, warmup=2))
        dev = x.device
        pool = [torch.cuda.Stream(device=dev) for _ in range(max(int(n_candidates), self.n))]
        cur = torch.cuda.current_stream(dev)
        for st in pool:
            st.wait_stream(cur)
        self.probe = {}
        best, best_rate = tuple(range(self.n)), -1.0
        if probe_steps > 0 and len(pool) > self.n:
            for cand in itertools.combinations(range(len(pool)), self.n):
                for rep in range(2):                                   # the first pass warms the set up
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for k in range(int(probe_steps)):
                        with torch.cuda.stream(pool[cand[k % self.n]]):
                            self.slots[k % self.n].graph.replay()
                    torch.cuda.synchronize(dev)
                    rate = probe_steps / (time.perf_counter() - t0)
                self.probe[cand] = rate
                if rate > best_rate:
                    best, best_rate = cand, rate
        self._pool = pool
        self.streams = [pool[i] for i in best]
        self._next = 0

    def reprobe(self, submit_fn, steps=40, top=6):
        """Pick the stream set again with the caller's OWN submit path (e.g. `lambda k: fl.submit_u8(...)`, whose host-to-device copies bring the copy engines'
        queues into play: a set that overlaps plain replays need not overlap fed ones): the `top` best sets of the constructor's probe, `steps` submits each."""
        import time
        if not self.probe:
            return
        dev = self.slots[0].x.device
        cands = sorted(self.probe, key=self.probe.get, reverse=True)[:int(top)]
        best, best_rate, rates = None, -1.0, {}
        for cand in cands:
            self.streams = [self._pool[i] for i in cand]
            for rep in range(2):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for k in range(int(steps)):
                    submit_fn(k)
                torch.cuda.synchronize(dev)
                rate = steps / (time.perf_counter() - t0)
            rates[cand] = rate
            if rate > best_rate:
                best, best_rate = cand, rate
        self.streams = [self._pool[i] for i in best]
        return rates

    def submit(self, x=None):
        slot = self._next
        self._next = (slot + 1) % self.n
        with torch.cuda.stream(self.streams[slot]):
            if x is not None:
                self.slots[slot].x.copy_(x, non_blocking=True)
            self.slots[slot].graph.replay()
        return slot, self.slots[slot].out

    def submit_u8(self, img_u8_host, means, im_scale=1.0):
        """The same from a uint8 HWC image in (pinned) host memory -- forward.py:85-94's cv.imread -> img_preprocessing -> model: the H2D copy (1.8 MB for
        600 x 1000), frcnn_preprocess_u8 into the slot's captured input and the graph replay all go to the SLOT's stream, so one image's copy and preprocessing
        run under the other slot's forward without a copy stream or cross-stream events."""
        slot = self._next
        self._next = (slot + 1) % self.n
        cf = self.slots[slot]
        rt = cf.model.rt
        if getattr(self, "_u8", None) is None:
            self._u8 = [None] * self.n
        with torch.cuda.stream(self.streams[slot]):
            if self._u8[slot] is None or tuple(self._u8[slot].shape) != tuple(img_u8_host.shape):
                self._u8[slot] = torch.empty(tuple(img_u8_host.shape), dtype=torch.uint8, device=cf.x.device)
            self._u8[slot].copy_(img_u8_host, non_blocking=True)
            rt.preprocess_u8(self._u8[slot], means, im_scale, (int(cf.x.shape[2]), int(cf.x.shape[3])), out=cf.x)
            cf.graph.replay()
        return slot, cf.out

    def wait(self, slot=None):
        for i in (range(self.n) if slot is None else (slot,)):
            self.streams[i].synchronize()
