from .dataset import WaferCollateLoader, WaferLoader, WaferMapDataset  # noqa: F401
from .store import WaferStore  # noqa: F401
from .ingest import convert_pickle, read_wafer_pickle  # noqa: F401
