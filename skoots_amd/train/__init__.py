"""Config-5 training step on the HIP kernels (reference: skoots/train/)."""
from .engine import TrainUNet, TrainStep, fused_loss, sync_gradients, train_step  # noqa: F401
from .loss import tversky  # noqa: F401
from .sigma import Sigma, init_sigma  # noqa: F401
