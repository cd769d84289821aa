"""vgsim_amd — MI355X-native forward epidemic simulation engine behind VGsim's ``Simulator`` API."""
from ._interface import Simulator
from ._model import BirthDeathModel

__all__ = ["Simulator", "BirthDeathModel"]
