"""Import name of the MI355X-native dense-retrieval package.

The sources live in ``lean-explore_amd/`` (the layout the build contract names); a hyphen is
not importable, so this two-line package extends its ``__path__`` to that directory. Every
submodule (``lean_explore_amd.native``, ``.index``, ``.faiss_compat``, ``.search`` ...) is
loaded from there.
"""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                              "lean-explore_amd"))

from ._version import __version__  # noqa: E402,F401
