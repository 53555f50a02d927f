from .capnet import CapNet  # noqa: F401
