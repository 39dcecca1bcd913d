"""Top-level alias so the reference's ``import model_content_extractor`` / ``from model_content_extractor import *`` resolves to the MI355X
implementation: put this directory (and the repository root) on PYTHONPATH.  See INTEGRATION.md."""
import importlib as _il

_m = _il.import_module('single-image-super-resolution_amd.model_content_extractor')
globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
