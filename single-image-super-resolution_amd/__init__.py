"""MI355X-native (gfx950) SRGAN training hot path with the module interface of
keyber/Single-Image-Super-Resolution.  See DESIGN.md / INTEGRATION.md at the repository root.

The package directory name contains hyphens, so import it with
``importlib.import_module("single-image-super-resolution_amd")`` (or put ``dropin/`` on
``sys.path`` to get ``model_generator`` / ``model_discriminator`` / ``model_content_extractor`` /
``utils`` under the reference's own module names).
"""
from . import _lib  # noqa: F401


def install(fused_adam=False):
    """Register this package's modules under the reference's top-level module names, so the
    reference's ``train.py`` / ``config.py`` import them unchanged.  ``fused_adam=True`` also points
    ``torch.optim.Adam`` (what config.py:293-294 instantiates) at the fused multi-tensor Adam of ``optim.py``."""
    import importlib
    import sys
    if fused_adam:
        import torch
        torch.optim.Adam = importlib.import_module('.optim', __name__).Adam
    for name in ('model_generator', 'model_generator_progressive', 'model_discriminator',
                 'model_content_extractor', 'utils'):
        try:
            sys.modules[name] = importlib.import_module('.' + name, __name__)
        except ModuleNotFoundError:
            pass
