"""MI355X-native FlashFFTConv.  Same public surface as the reference package
(/root/reference/flashfftconv/__init__.py:1-2)."""
from .conv import FlashFFTConv
from .depthwise_1d import FlashDepthWiseConv1d
from .sparse_conv import PartialFFTConv, FrequencySparseFFTConv
from .graphs import GraphedStep
from .hyena import FlashHyenaOp, FlashHyenaMixer, gated_conv_from_slices, project_in, project_out

FlashDepthwiseConv1d = FlashDepthWiseConv1d  # README spelling of the reference
__all__ = ["FlashFFTConv", "FlashDepthWiseConv1d", "FlashDepthwiseConv1d", "PartialFFTConv", "FrequencySparseFFTConv",
           "GraphedStep", "FlashHyenaOp", "FlashHyenaMixer", "gated_conv_from_slices", "project_in", "project_out"]
