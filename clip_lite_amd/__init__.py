"""Import shim: the package sources live in ``clip-lite_amd/`` (the directory name the project layout fixes, which is
not a valid Python identifier). ``import clip_lite_amd.<module>`` resolves to ``clip-lite_amd/<module>.py``."""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "clip-lite_amd"))
__version__ = "0.1.0"
