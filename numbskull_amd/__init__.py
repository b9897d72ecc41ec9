"""MI355X-native Gibbs sampling and weight learning for DeepDive-style factor graphs."""

from .version import __version__
