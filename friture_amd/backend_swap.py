"""The import swap of INTEGRATION.md §2 as code: after `install()` the reference's own import lines
(`from friture.audioproc import audioproc`, `from friture.octavefilters import Octave_Filters, NOCTAVE`,
`from friture.filter import octave_filter_bank_decimation, ...`, `from friture.signal.correlation import
generalized_cross_correlation`, ...) resolve to this package's classes, i.e. to libfriture_hip.so.

A Friture maintainer calls it once, before the widgets are imported (top of friture/analyzer.py:main); the
module list is the set of reference modules SURVEY.md §8b names as the drop-in boundary.  `uninstall()` puts
back whatever was there (tests)."""
from __future__ import annotations

import importlib
import sys

SWAPPED = ("audioproc", "octavefilters", "filter", "ringbuffer",
           "signal.correlation", "signal.decimate", "signal.lfilter", "signal.exp_smoothing",
           "signal.frequency_resampler", "signal.online_linear_2D_resampler", "signal.scipy_resample",
           "signal.color_tranform", "signal.transform_pipeline")

_saved: dict = {}
_saved_attr: dict = {}          # (parent module name, leaf) -> the attribute install() replaced (_MISSING when there was none)
_MISSING = object()


def install(target_package: str = "friture") -> None:
    """Make `<target_package>.<name>` resolve to `friture_amd.<name>` for every module of the boundary."""
    for name in SWAPPED:
        mod = importlib.import_module("friture_amd." + name)
        key = target_package + "." + name
        if key not in _saved:
            _saved[key] = sys.modules.get(key)
        sys.modules[key] = mod
        # `import pkg.sub` binds the submodule as an attribute of its parent: keep attribute access consistent
        parent_name, _, leaf = key.rpartition(".")
        parent = sys.modules.get(parent_name)
        if parent is not None:
            if (parent_name, leaf) not in _saved_attr:
                _saved_attr[(parent_name, leaf)] = getattr(parent, leaf, _MISSING)
            setattr(parent, leaf, mod)


def uninstall() -> None:
    for key, old in _saved.items():
        if old is None:
            sys.modules.pop(key, None)
        else:
            sys.modules[key] = old
    _saved.clear()
    # the parent packages' attributes too: `import friture.signal.correlation as m` resolves through them
    for (parent_name, leaf), old in _saved_attr.items():
        parent = sys.modules.get(parent_name)
        if parent is None:
            continue
        if old is _MISSING:
            if hasattr(parent, leaf):
                delattr(parent, leaf)
        else:
            setattr(parent, leaf, old)
    _saved_attr.clear()
