"""GEMM kernel selections made offline on an MI355X, applied at start-up.

The Linear / GRU-projection GEMMs of the three models have a handful of shapes (8000 rows x 512 / 1536 / 2048 / 9072);
hipBLASLt's default heuristic is not the fastest kernel for several of them (tools/bench_tunable.sh: the table under
``tuned/`` was produced by PyTorch's TunableOp timing every hipBLASLt / rocBLAS solution on the card).  The table is
keyed by the library versions it was tuned with (its ``Validator`` rows): on any other stack PyTorch discards it and
this module switches the lookup off again, so a stale table never selects a kernel.  Nothing is tuned at run time
(``SELD_TUNED_GEMMS=tune`` is the developer switch that does, writing ``SELD_TUNED_GEMMS_OUT``)."""
import os
from pathlib import Path

import torch

TABLE = Path(__file__).resolve().parent / "tuned" / "gemm_gfx950.csv"
state = {"mode": "off", "entries": 0}


def enable(device, wanted=True):
    """Called once per process by trainer.prepare_model_for_device."""
    mode = os.environ.get("SELD_TUNED_GEMMS", "1" if wanted else "0")
    if device.type != "cuda" or mode == "0" or state["mode"] != "off":
        return state
    tunable = torch.cuda.tunable
    if mode == "tune":
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_filename(os.environ.get("SELD_TUNED_GEMMS_OUT", "gemm_tuned.csv"))
        state["mode"] = "tune"
        return state
    if not TABLE.exists():
        return state
    tunable.enable(True)
    tunable.tuning_enable(False)
    try:
        tunable.write_file_on_exit(False)
    except AttributeError:
        pass
    ok = tunable.read_file(str(TABLE))
    if not ok:
        tunable.enable(False)
        state["mode"] = "stale table ignored"
        return state
    state["mode"] = "table"
    state["entries"] = len(tunable.get_results())
    return state
