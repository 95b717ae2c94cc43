"""neural_marionette_amd — MI355X-native hot path of Neural Marionette.

    from neural_marionette_amd import NeuralMarionette      # drop-in for model.neural_marionette

The computation lives in libnm355.so (hand-written HIP for gfx950, C ABI in include/nm355.h);
this package holds the host-side mirror of the reference's nn.Module surface, the host
skeleton builder and synthetic data generators.  No CPU / PyTorch fallback exists.
"""
from .spec import HotPathOptions, param_spec, param_count, DETECTOR_LOSS_KEYS  # noqa: F401
from .modules import NeuralMarionette, KyptDetector, HSVRNNBVH  # noqa: F401

__all__ = ["NeuralMarionette", "KyptDetector", "HSVRNNBVH", "HotPathOptions", "param_spec", "param_count"]
