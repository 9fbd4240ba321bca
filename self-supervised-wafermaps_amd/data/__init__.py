from .dataset import WaferLoader, WaferMapDataset  # noqa: F401
from .store import WaferStore  # noqa: F401
