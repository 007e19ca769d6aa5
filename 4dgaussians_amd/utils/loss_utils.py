"""utils/loss_utils.py of the reference, hot-path part: l1_loss (:20-21) as a fused HIP operator."""
from ..fused import l1_loss  # noqa: F401
