from .store import WaferStore  # noqa: F401
