"""HIP-graph form of a FlashFFTConv training step (short sequences).

At fft sizes <= 2048 a forward + backward of FlashFFTConv is bound by the host, not by the GPU: an `autograd.Function` round trip
costs ~25 us, the hand-over to the autograd engine's thread ~50 us, against 30 - 50 us of kernels at B=16 H=768
(profiles/r04_host_overhead.txt).  Every launch of the library goes to torch's current stream, allocates only through torch's
caching allocator and never synchronises with the host, so the whole step -- forward, backward, every gradient -- captures
into ONE graph and replays with a single `hipGraphLaunch`:

    step = conv.graphed_step(u, k, dout)                   # or (u, k, dout, pregate, postgate); captures once (warm-up + capture)
    y, du, dk = step(u_new, k_new, dout_new)               # copies into the static inputs, one graph launch
    step.u.copy_(...); step.replay()                       # zero-copy form: write the static tensors yourself
    step.y, step.du, step.dk (, step.dpregate, step.dpostgate)

The reference has no equivalent (its README.md:224-231 table is per-call timing of eager launches); this is the MI355X-side answer
to "launch-bound inner loops belong in hipGraphs".  `torch.cuda.make_graphed_callables(conv, (u, k))` also works on the module
(separate forward / backward graphs inside autograd) and is what a model that keeps autograd around the convolution should use;
it still pays autograd's per-call cost, the whole-step graph does not."""
import torch


class GraphedStep:
    """One captured forward + backward of `conv` on static tensors of the given shapes."""

    def __init__(self, conv, u, k, dout, pregate=None, postgate=None, warmup=3, pool=None):
        if (pregate is None) != (postgate is None):
            raise RuntimeError("graphed_step: pregate and postgate come together")
        if not u.is_cuda:
            raise RuntimeError("graphed_step: CUDA/HIP tensors only")
        self.conv = conv
        self.gated = pregate is not None
        # static inputs: private copies, so that the caller's tensors can be freed / reused
        self.u = u.detach().clone().requires_grad_(True)
        self.k = k.detach().clone().requires_grad_(True)
        self.dout = dout.detach().clone()
        self.pregate = pregate.detach().clone().requires_grad_(True) if self.gated else None
        self.postgate = postgate.detach().clone().requires_grad_(True) if self.gated else None
        self._leaves = [self.u, self.k] + ([self.pregate, self.postgate] if self.gated else [])
        was_training = conv.training
        conv.train()
        try:
            # warm-up on a side stream (plan creation, allocator growth, LDS attributes: none of that may happen under capture)
            s = torch.cuda.Stream(device=u.device)
            s.wait_stream(torch.cuda.current_stream(u.device))
            with torch.cuda.stream(s):
                for _ in range(max(1, warmup)):
                    self._eager()
            torch.cuda.current_stream(u.device).wait_stream(s)
            # whether the spectra are kept is decided HERE, by the budget test of the last warm-up step, and pinned for the capture: under
            # capture hipMemGetInfo is off limits and an allocator OOM would poison the graph instead of falling back.  An OOM
            # during the capture itself is fatal for the graph (RuntimeError from torch.cuda.graph), as for any captured region.
            keep_mode = conv.save_spectrum
            if keep_mode is True:
                from . import conv as _C
                n = ((u.shape[0] + 1) // 2) * u.shape[1] * conv.seqlen * (8 if self.gated else 4)
                conv.save_spectrum = "always" if _C._spectrum_budget_ok(n, u.device, True) else False
            try:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, pool=pool):
                    y, grads = self._eager()
            finally:
                conv.save_spectrum = keep_mode
        finally:
            conv.train(was_training)
        self.y = y
        self.du, self.dk = grads[0], grads[1]
        self.dpregate, self.dpostgate = (grads[2], grads[3]) if self.gated else (None, None)

    def _eager(self):
        args = (self.u, self.k) + ((self.pregate, self.postgate) if self.gated else ())
        y = self.conv(*args)
        grads = torch.autograd.grad(y, self._leaves, self.dout)
        return y.detach(), grads

    def replay(self):
        """one graph launch on the current stream; results in self.y / self.du / self.dk (/ self.dpregate / self.dpostgate)"""
        self.graph.replay()

    def _same(self, name, static, new):
        if new is None or tuple(new.shape) != tuple(static.shape) or new.dtype != static.dtype:
            raise RuntimeError(f"graphed step: {name} must be {tuple(static.shape)} {static.dtype} (the captured shapes), got "
                               f"{None if new is None else (tuple(new.shape), new.dtype)}")      # (copy_ would broadcast silently: ADVICE r05)

    def __call__(self, u, k, dout, pregate=None, postgate=None):
        self._same("u", self.u, u); self._same("k", self.k, k); self._same("dout", self.dout, dout)
        if self.gated:
            self._same("pregate", self.pregate, pregate); self._same("postgate", self.postgate, postgate)
        self.u.detach().copy_(u); self.k.detach().copy_(k); self.dout.copy_(dout)
        if self.gated:
            self.pregate.detach().copy_(pregate); self.postgate.detach().copy_(postgate)
        self.graph.replay()
        if self.gated:
            return self.y, self.du, self.dk, self.dpregate, self.dpostgate
        return self.y, self.du, self.dk
