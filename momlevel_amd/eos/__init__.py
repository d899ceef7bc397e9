""" momlevel_amd - equation of state module (mirrors src/momlevel/eos/__init__.py)

Any module here exposing ``density(T, S, p)`` is reachable through
``util.eos_func_from_str`` -- the reference's string-keyed EOS seam
(src/momlevel/util.py:227-249).
"""

from . import linear
from . import wright
