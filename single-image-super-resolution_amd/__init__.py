"""MI355X-native (gfx950) SRGAN training hot path with the module interface of
keyber/Single-Image-Super-Resolution.  See DESIGN.md / INTEGRATION.md at the repository root.

The package directory name contains hyphens, so import it with
``importlib.import_module("single-image-super-resolution_amd")`` (or put ``dropin/`` on
``sys.path`` to get ``model_generator`` / ``model_discriminator`` / ``model_content_extractor`` under the
reference's own module names; ``install()`` additionally patches ``utils.lr_from_hr``).
"""
from . import _lib  # noqa: F401


def install(fused_adam=False):
    """Register this package's model modules under the reference's top-level module names, so the
    reference's ``train.py`` / ``config.py`` import them unchanged, and point ``utils.lr_from_hr`` -- the one hot
    function of the reference's ``utils`` module (utils.py:22-31; called at train.py:46,96, config.py:272) -- at the
    HIP kernel.  The reference's own ``utils`` module is NOT shadowed: its plotting / checkpoint helpers
    (``save_and_show``, ``save_curr_vis``, ``SamplerRange`` ... train.py:15,36, config.py:250) stay the reference's;
    only when no ``utils`` module is importable at all (this repository's own tests) is the package's ``utils``
    registered under that name.  ``fused_adam=True`` also points ``torch.optim.Adam`` (what config.py:293-294
    instantiates) at the fused multi-tensor Adam of ``optim.py``."""
    import importlib
    import sys
    if fused_adam:
        import torch
        torch.optim.Adam = importlib.import_module('.optim', __name__).Adam
    for name in ('model_generator', 'model_generator_progressive', 'model_discriminator', 'model_content_extractor'):
        sys.modules[name] = importlib.import_module('.' + name, __name__)
    ours = importlib.import_module('.utils', __name__)
    try:
        ref_utils = importlib.import_module('utils')          # the reference's utils.py, when it is on the path
    except ImportError:
        sys.modules['utils'] = ours
        return
    if ref_utils is not ours:
        ref_utils.lr_from_hr = ours.lr_from_hr
        ref_utils._subsampling_interpolation = ours._subsampling_interpolation
