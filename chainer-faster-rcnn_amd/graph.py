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
