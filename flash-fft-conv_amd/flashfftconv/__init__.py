"""MI355X-native FlashFFTConv.  Same public surface as the reference package
(/root/reference/flashfftconv/__init__.py:1-2)."""
from .conv import FlashFFTConv
from .depthwise_1d import FlashDepthWiseConv1d
from .sparse_conv import PartialFFTConv, FrequencySparseFFTConv
from .hyena import FlashHyenaOp, gated_conv_from_slices

FlashDepthwiseConv1d = FlashDepthWiseConv1d  # README spelling of the reference
__all__ = ["FlashFFTConv", "FlashDepthWiseConv1d", "FlashDepthwiseConv1d", "PartialFFTConv", "FrequencySparseFFTConv",
           "FlashHyenaOp", "gated_conv_from_slices"]
