"""slak_amd -- MI355X (gfx950) native hot path of VITA-Group/SLaK: the decomposed large-kernel
depthwise conv (forward / data-grad / weight-grad) and the Masking prune-grow-apply step, as
hand-written HIP behind the reference's own operator surface.  See DESIGN.md / INTEGRATION.md."""
from .depthwise_conv2d_implicit_gemm import DepthWiseConv2dImplicitGEMM  # noqa: F401

__all__ = ["DepthWiseConv2dImplicitGEMM"]
