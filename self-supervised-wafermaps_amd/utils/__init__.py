from . import benchmarking, debug, scheduler  # noqa: F401
