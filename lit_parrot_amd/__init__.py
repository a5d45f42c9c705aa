"""Importable alias of the ``lit-parrot_amd/`` package directory.

The package directory carries the reference's name (``lit-parrot`` + ``_amd``), which is not a valid Python
identifier; this stub makes it importable as ``lit_parrot_amd`` by pointing ``__path__`` at it.
"""
from pathlib import Path as _Path

__path__ = [str(_Path(__file__).resolve().parent.parent / "lit-parrot_amd")]

from lit_parrot_amd._api import *  # noqa: E402,F401,F403
from lit_parrot_amd._api import __all__  # noqa: E402,F401
