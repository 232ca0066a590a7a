"""Importable alias of the package directory ``svt-av1-psyex_amd/`` (a hyphen cannot be imported)."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "svt-av1-psyex_amd"))
