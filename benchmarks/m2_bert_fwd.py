"""End-to-end caller bench (second half): an M2-BERT encoder forward, the port of the
reference's examples/bert/benchmark_fwd.py to this package.

The sequence mixer is the reference's `MonarchMixerSequenceMixingFlashFFTConv`
(examples/bert/monarch_mixer_sequence_mixer_flashfftconv.py:118-175) in the configuration the published M2-BERT checkpoints
use: bidirectional (the filter has 2 L taps, k = pad(k_fwd, (0, L)) + pad(flip(k_rev), (L, 0)), so fft size = 2 L and the
kernel fills it), residual long convolution (y = x2 * conv(x1 * v, k) + conv(v, k2)) and inference mode (the two filters are
parameters; :57-64).  Random weights (no network: no checkpoint).  The encoder around it is LayerNorm -> mixer -> +res ->
LayerNorm -> GLU MLP (4x) -> +res; the reference's MLP is a block-diagonal Monarch GLU (bert_layers.py), here a dense GLU of the
same shapes: it is not on this package's path and is the same code in all three forms.
   fused    gated_conv_from_slices (x1 / x2 / v read in place, gates inside the kernel) + FlashFFTConv(v, k2)
   dropin   the reference caller code verbatim on this package's modules (:125-170)
   torch    nn.Conv1d + torch.fft (the non-flash mixer, monarch_mixer_sequence_mixer.py)
Prints one JSON line per (config, form): ms per forward, tokens/ms, seqs/s (the three numbers the reference prints) and the
relative difference of the output to the torch form.
usage: python benchmarks/m2_bert_fwd.py [base-128 base-2k base-8k base-32k]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "flash-fft-conv_amd"), ROOT]
import torch
import torch.nn as nn
import torch.nn.functional as F
from flashfftconv import FlashFFTConv, FlashDepthWiseConv1d
from flashfftconv.hyena import gated_conv_from_slices, project_in, project_out

# name: (d_model, n_layer, seq_len, batch)   -- M2-BERT-base 80M (12 layers x 768) at its four published context lengths
CONFIGS = {"base-128": (768, 12, 128, 32), "base-2k": (768, 12, 2048, 8), "base-8k": (768, 12, 8192, 4),
           "base-32k": (768, 12, 32768, 2)}


def bidirectional_filter(d_model, L):
    """k = pad(k_fwd, (0, L)) + pad(flip(k_rev), (L, 0)) (:141-142) with decaying random k_fwd / k_rev"""
    t = torch.linspace(0, 1, L)[None]
    decay = torch.exp(-t * torch.linspace(2.0, 12.0, d_model)[:, None])
    k, k_rev = (torch.randn(d_model, L) * 0.02 * decay for _ in range(2))
    return F.pad(k, (0, L)) + F.pad(k_rev.flip(-1), (L, 0))


class SequenceMixer(nn.Module):
    def __init__(self, d_model, l_max, form, dtype):
        super().__init__()
        self.d_model, self.form = d_model, form
        self.filter = nn.Parameter(bidirectional_filter(d_model, l_max))
        self.filter2 = nn.Parameter(bidirectional_filter(d_model, l_max))
        self.in_linear = nn.Linear(d_model, 3 * d_model)
        self.out_linear = nn.Linear(d_model, d_model)
        self.conv1d = nn.Conv1d(3 * d_model, 3 * d_model, 3, groups=3 * d_model, padding=2)
        if form != "torch":
            self.short_filter = FlashDepthWiseConv1d(3 * d_model, 3, padding=1, weights=self.conv1d.weight,
                                                     bias=self.conv1d.bias, dtype=dtype)
            self.flashfftconv = FlashFFTConv(2 * l_max, dtype=dtype)

    def forward(self, u):
        B, L, H = u.shape
        k, k2 = self.filter.float(), self.filter2.float()
        if self.form == "fused":      # projections as batched GEMMs on transposed views: no layout copy on either side
            uc = self.short_filter(project_in(self.in_linear.weight, u))
            y = gated_conv_from_slices(self.flashfftconv, uc, k)  # x2 * conv(x1 * v, k), slices read in place
            y = y + self.flashfftconv(uc[:, 2 * H:].contiguous(), k2)
            return project_out(self.out_linear.weight, self.out_linear.bias, y)
        u = u.transpose(-1, -2)
        x1x2v = (self.in_linear.weight @ u).contiguous()          # the reference drops the in_linear bias (:124-125)
        if self.form == "dropin":                               # :128-170, verbatim
            x1x2v = self.short_filter(x1x2v)
            x1, x2, v = x1x2v.split(self.d_model, dim=1)
            x1v = x1 * v
            x1v = x1v.contiguous()
            y = self.flashfftconv(x1v, k)
            v = v.contiguous()
            yu = self.flashfftconv(v, k2)
            y = y * x2
            y = y + yu
        else:
            x1x2v = F.conv1d(x1x2v, self.conv1d.weight, self.conv1d.bias, padding=1, groups=3 * H)
            x1, x2, v = x1x2v.split(self.d_model, dim=1)
            n = 2 * L
            fc = lambda x, kk: torch.fft.irfft(torch.fft.rfft(x.float(), n=n) * torch.fft.rfft(kk, n=n), n=n)[..., :L].to(x.dtype)
            y = fc(x1 * v, k) * x2 + fc(v, k2)
        return self.out_linear(y.transpose(-1, -2))


class Encoder(nn.Module):
    def __init__(self, d_model, n_layer, l_max, form, dtype):
        super().__init__()
        self.layers = nn.ModuleList()
        for _ in range(n_layer):
            self.layers.append(nn.ModuleDict(dict(
                n1=nn.LayerNorm(d_model), mixer=SequenceMixer(d_model, l_max, form, dtype), n2=nn.LayerNorm(d_model),
                glu=nn.Linear(d_model, 2 * 4 * d_model), out=nn.Linear(4 * d_model, d_model))))

    def forward(self, x):
        for l in self.layers:
            x = l["n1"](x + l["mixer"](x))                         # post-LN residual blocks, as BERT
            g, v = l["glu"](x).chunk(2, dim=-1)
            x = l["n2"](x + l["out"](F.gelu(g) * v))
        return x


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


def run(name, dtype=torch.bfloat16):
    d_model, n_layer, L, B = CONFIGS[name]
    torch.manual_seed(1)
    u = torch.randn(B, L, d_model, device="cuda").to(dtype)
    outs = {}
    for form in ("torch", "dropin", "fused"):
        torch.manual_seed(0)
        model = Encoder(d_model, n_layer, L, form, dtype).cuda().to(dtype).eval()
        with torch.no_grad():
            y = model(u)
            ms = min(ev_time(lambda: model(u), 10) for _ in range(3))      # best of 3: a shared box shows 50 % outliers
        outs[form] = y.float()
        diff = ((outs[form] - outs["torch"]).norm() / outs["torch"].norm()).item()
        print(json.dumps({"model": f"m2-bert-{name}", "d_model": d_model, "n_layer": n_layer, "seqlen": L, "batch": B,
                          "fft_size": 2 * L, "form": form, "dtype": str(dtype).split(".")[-1], "ms": round(ms, 3),
                          "tokens_per_ms": round(B * L / ms, 1), "seqs_per_s": round(B / (ms * 1e-3), 2),
                          "rel_diff_vs_torch": round(diff, 5)}), flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(CONFIGS)):
        run(n)
