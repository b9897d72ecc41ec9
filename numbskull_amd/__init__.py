"""MI355X-native Gibbs sampling and weight learning for DeepDive-style factor graphs.

Public surface = the reference package's (numbskull/__init__.py): ``NumbSkull``, ``main``,
``__version__``; submodules ``numbskull``, ``factorgraph``, ``inference`` (factor table),
``numbskulltypes``, ``dataloading``, ``timer``.
"""

from .version import __version__
from .numbskull import NumbSkull
from .numbskull import main

__all__ = ('numbskull', 'factorgraph', 'timer')
