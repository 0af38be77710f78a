"""Import shim: the product package lives in ``image-super-resolution-2_amd/`` (a directory
name Python cannot import directly because of the hyphens).  ``import isr2_amd.<module>``
resolves to ``image-super-resolution-2_amd/<module>.py``."""
import os as _os

_PKG_DIR = _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..",
                                           "image-super-resolution-2_amd"))
__path__.insert(0, _PKG_DIR)
PKG_DIR = _PKG_DIR
