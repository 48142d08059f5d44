"""End-to-end caller bench: a HyenaDNA backbone forward, the port of the reference's
examples/hyena-dna/benchmark_flash_dna_fwd.py to this package.

The model is the reference's `HyenaDNAModel` backbone (examples/hyena-dna/hyenadna_flashfftconv.py: token embedding ->
n_layer x [LayerNorm -> HyenaOperator -> +res -> LayerNorm -> MLP(4x, GELU) -> +res] -> LayerNorm) with random weights
(no network: no HuggingFace checkpoint) and, like the reference's flash model, a static long filter (`HyenaFilter.filter`
returns the k_ones buffer, :190-201).  The Hyena operator comes in three forms:
   fused    this package's FlashHyenaMixer (projections as batched GEMMs on transposed views, short conv + gated FFT conv with
            the slices read in place: no layout copy, no elementwise kernel)
   dropin   the reference CALLER code verbatim on this package's modules (FlashDepthWiseConv1d, x1*v, .contiguous(),
            FlashFFTConv, *x2): hyenadna_flashfftconv.py:269-289 -- the drop-in claim on a real caller
   torch    the reference's non-flash path: nn.Conv1d + torch.fft (hyenadna_standalone.py fftconv)
Prints one JSON line per (config, form): ms per forward, tokens/ms, seqs/s (the reference prints the same three numbers),
and the relative difference of the output to the torch form.
The config `hyena-pile-4k` is the port of the reference's third caller bench, examples/hyena/benchmark_fwd.py (a hydra /
lightning harness around `model.model.backbone(input_ids)`, :665-680) on its sample config experiment/pile/hyena-flashfft.yaml:
the 153M Hyena LM backbone (d_model 864, 18 layers, d_inner = 2 d_model, l_max 4096, batch 8, fp16) in inference mode (the
long filter is loaded, not generated: examples/hyena/README.md), same three printed numbers.
usage: python benchmarks/hyena_dna_fwd.py [tiny-16k small-32k medium-160k large-1m hyena-pile-4k]"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
import torch
import torch.nn as nn
import torch.nn.functional as F
from flashfftconv import FlashFFTConv, FlashDepthWiseConv1d, FlashHyenaMixer

# name: (d_model, n_layer, max_length, batch[, d_inner / d_model, dtype])   -- sizes of the published HyenaDNA checkpoints
# (huggingface.py:160-175) and the Hyena LM of examples/hyena/configs/experiment/pile/hyena.yaml
CONFIGS = {"tiny-16k": (128, 2, 16384, 4), "small-32k": (256, 4, 32768, 4), "medium-160k": (256, 8, 131072, 4),
           "medium-450k": (256, 8, 450000, 2), "large-1m": (256, 8, 1000000, 2),
           "hyena-pile-4k": (864, 18, 4096, 8, 2, torch.float16)}


def fft_size_for(L):
    n = 256
    while n < 2 * L:
        n *= 2
    return n


def static_filter(d_model, L):
    """HyenaFilter.__init__ (:186-199): randn * 0.001 under the exponential-decay window"""
    t = torch.linspace(0, 1, L)[None, :, None]
    deltas = torch.linspace(math.log(1e-2) / 1.5, math.log(1e-2) / 0.3, d_model)[None, None]
    k = torch.randn(1, L, d_model) * 0.001 * (torch.exp(-t * deltas.abs()) + 0.05)
    return k.transpose(-1, -2).squeeze(0).contiguous()


class HyenaOperator(nn.Module):
    def __init__(self, d_model, l_max, form, dtype):
        super().__init__()
        self.d_model, self.l_max, self.form, self.dtype = d_model, l_max, form, dtype
        self.in_proj = nn.Linear(d_model, 3 * d_model)
        self.out_proj = nn.Linear(d_model, d_model)
        self.short_filter = nn.Conv1d(3 * d_model, 3 * d_model, 3, padding=2, groups=3 * d_model)
        self.register_buffer("k", static_filter(d_model, l_max))
        n = fft_size_for(l_max)
        if form == "fused":
            self.mixer = FlashHyenaMixer(d_model, n, self.in_proj, self.out_proj, self.short_filter.weight, self.short_filter.bias, dtype=dtype)
        elif form == "dropin":
            self.flash_short_filter = FlashDepthWiseConv1d(3 * d_model, 3, padding=1, weights=self.short_filter.weight,
                                                           bias=self.short_filter.bias, dtype=dtype)
            self.flashfftconv = FlashFFTConv(n, dtype=dtype)

    def forward(self, u):
        l = u.size(-2)
        if self.form == "fused":      # projections as batched GEMMs on transposed views, gates read in place: no layout copy
            return self.mixer(u, self.k[:, :l])
        u = u.transpose(-1, -2)
        u = (self.in_proj.weight @ u).contiguous()        # the reference drops the in_proj bias here too (:270)
        k = self.k[:, :l]
        if self.form == "dropin":                       # hyenadna_flashfftconv.py:272-284, verbatim
            uc = self.flash_short_filter(u)[..., :l]
            x1, x2, v = uc.split(self.d_model, dim=1)
            x1v = x1 * v
            x1v = x1v.contiguous()
            y = self.flashfftconv(x1v, k)
            y = y * x2
        else:                                             # hyenadna_standalone.py: short conv + torch.fft long conv
            uc = F.conv1d(u, self.short_filter.weight.to(u.dtype), self.short_filter.bias.to(u.dtype), padding=1, groups=u.shape[1])[..., :l]
            x1, x2, v = uc.split(self.d_model, dim=1)
            n = 2 * l
            y = torch.fft.irfft(torch.fft.rfft((x1 * v).float(), n=n) * torch.fft.rfft(k.float(), n=n), n=n)[..., :l].to(u.dtype)
            y = y * x2
        return self.out_proj(y.transpose(-1, -2))


class Backbone(nn.Module):
    def __init__(self, d_model, n_layer, l_max, form, dtype, vocab=16, inner=4):
        super().__init__()
        self.emb = nn.Embedding(vocab, d_model)
        self.layers = nn.ModuleList()
        for _ in range(n_layer):
            self.layers.append(nn.ModuleDict(dict(
                n1=nn.LayerNorm(d_model), mixer=HyenaOperator(d_model, l_max, form, dtype), n2=nn.LayerNorm(d_model),
                fc1=nn.Linear(d_model, inner * d_model), fc2=nn.Linear(inner * d_model, d_model))))
        self.ln_f = nn.LayerNorm(d_model)

    def forward(self, ids):
        x = self.emb(ids)
        for l in self.layers:
            x = x + l["mixer"](l["n1"](x))
            x = x + l["fc2"](F.gelu(l["fc1"](l["n2"](x)), approximate="tanh"))
        return self.ln_f(x)


def ev_time(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


FORMS = ("torch", "dropin", "fused")
TRAIN = False


def run(name, dtype=torch.bfloat16):
    d_model, n_layer, L, B = CONFIGS[name][:4]
    inner = CONFIGS[name][4] if len(CONFIGS[name]) > 4 else 4
    dtype = CONFIGS[name][5] if len(CONFIGS[name]) > 5 else dtype
    ids = torch.randint(0, 12, (B, L), device="cuda")
    outs = {}
    for form in FORMS:
        torch.manual_seed(0)
        model = Backbone(d_model, n_layer, L, form, dtype, inner=inner).cuda().to(dtype).eval()
        for m in model.modules():                    # the long filter stays fp32 (the reference passes fp32 k to FlashFFTConv)
            if isinstance(m, HyenaOperator):
                m.k = m.k.float()
        if TRAIN:       # --train: forward + backward of the backbone (training mode: the convolutions keep their spectra)
            model.train()
            params = [p for p in model.parameters() if p.requires_grad]
            for m in model.modules():                # the long filter is a learned quantity in training (HyenaFilter's MLP output)
                if isinstance(m, HyenaOperator):
                    m.k.requires_grad_(True); params.append(m.k)
            y = model(ids)
            dy = torch.randn_like(y) * 0.01

            def step():
                for p in params:
                    p.grad = None
                model(ids).backward(dy)
            ms = ev_time(step, 3 if L > 200000 else 10)
            y = y.detach()
        else:
            with torch.no_grad():
                y = model(ids)
                ms = ev_time(lambda: model(ids), 3 if L > 200000 else 10)
        outs[form] = y.float()
        diff = ((outs[form] - outs["torch"]).norm() / outs["torch"].norm()).item() if "torch" in outs else float("nan")
        print(json.dumps({"model": f"hyenadna-{name}", "d_model": d_model, "n_layer": n_layer, "seqlen": L, "batch": B,
                          "fft_size": fft_size_for(L), "form": form, "pass": "fwd+bwd" if TRAIN else "fwd", "dtype": str(dtype).split(".")[-1], "ms": round(ms, 3),
                          "tokens_per_ms": round(B * L / ms, 1), "seqs_per_s": round(B / (ms * 1e-3), 2),
                          "rel_diff_vs_torch": round(diff, 5)}), flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--train":
        TRAIN = True
        args = args[1:]
    if args and args[0].startswith("--forms="):      # e.g. --forms=dropin under rocprofv3 --stats: one form's kernels only
        FORMS = tuple(args[0][8:].split(","))
        args = args[1:]
    for n in (args or ["tiny-16k", "small-32k", "medium-160k", "large-1m"]):
        run(n)
