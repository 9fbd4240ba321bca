"""Record of every oracle comparison the GPU tests make: `parity(name, measured, bound)` asserts measured <= bound
(or >= for `higher=True`) and appends the pair to gpurun_out/parity_errors.jsonl; tools/parity_report.py turns the
file into profiles/r03_parity_errors.md.  The bounds written in the tests are set from these measurements (<= 2x the
measured value unless a test says why not)."""
import json
import os

_PATH = os.path.join("gpurun_out", "parity_errors.jsonl")


def parity(name: str, measured: float, bound: float, higher: bool = False, note: str = "") -> float:
    measured, bound = float(measured), float(bound)
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(_PATH, "a") as fh:
            fh.write(json.dumps({"name": name, "measured": measured, "bound": bound, "higher": higher, "note": note}) + "\n")
    except OSError:
        pass
    if higher:
        assert measured >= bound, f"{name}: measured {measured:.6g} < bound {bound:.6g}"
    else:
        assert measured <= bound, f"{name}: measured {measured:.6g} > bound {bound:.6g}"
    return measured
