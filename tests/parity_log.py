"""Record of every oracle comparison the GPU tests make: `parity(name, measured, bound)` asserts measured <= bound
(or >= for `higher=True`) and appends the pair to gpurun_out/parity_errors.jsonl; tools/parity_report.py turns the
file into profiles/r03_parity_errors.md.  The bounds written in the tests are set from these measurements (<= 2x the
measured value unless a test says why not)."""
import json
import os

_PATH = os.path.join("gpurun_out", "parity_errors.jsonl")


def _test_id() -> str:
    return os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]


def _write(rec: dict) -> None:
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(_PATH, "a") as fh:
            fh.write(json.dumps(rec) + "\n")
    except OSError:
        pass


def parity(name: str, measured: float, bound: float, higher: bool = False, note: str = "") -> float:
    measured, bound = float(measured), float(bound)
    _write({"name": name, "measured": measured, "bound": bound, "higher": higher, "note": note, "test": _test_id()})
    if higher:
        assert measured >= bound, f"{name}: measured {measured:.6g} < bound {bound:.6g}"
    else:
        assert measured <= bound, f"{name}: measured {measured:.6g} > bound {bound:.6g}"
    return measured


def tolerance_used(got, ref, rtol: float, atol: float) -> float:
    """max |got - ref| / (atol + rtol * |ref|): the fraction of an assert_close / allclose tolerance a comparison
    consumed (1 = at the limit; 0 = exact)."""
    import numpy as np
    import torch

    def arr(x):
        if isinstance(x, torch.Tensor):
            return x.detach().to("cpu", torch.float64).numpy()
        return np.asarray(x, dtype=np.float64)

    g, r = arr(got), arr(ref)
    if g.size == 0:
        return 0.0
    g, r = np.broadcast_arrays(g, r)
    fin = np.isfinite(g) & np.isfinite(r)
    if not fin.any():
        return 0.0
    den = atol + rtol * np.abs(r[fin])
    err = np.abs(g[fin] - r[fin])
    with np.errstate(divide="ignore", invalid="ignore"):
        frac = np.where(den > 0, err / den, np.where(err > 0, np.inf, 0.0))
    return float(frac.max())


def record_tolerance(kind: str, got, ref, rtol: float, atol: float) -> None:
    """One line per assert_close / allclose call of a GPU test (tests/conftest.py patches them in)."""
    try:
        used = tolerance_used(got, ref, rtol, atol)
    except Exception:  # non-numeric arguments: nothing to record
        return
    _write({"name": f"{kind}(rtol={rtol:g}, atol={atol:g})", "measured": used, "bound": 1.0, "higher": False,
            "note": "fraction of the tolerance used", "test": _test_id(), "generic": True})
