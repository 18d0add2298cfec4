"""The reference's entry script (main.py:15-18, 41-104) must run UNCHANGED against this package: every name it imports,
every keyword it passes, every Config attribute it reads and every key of the returned dicts it indexes, as extracted from
main.py's syntax tree by tests/golden/make_main_contract.py (main.py itself cannot be imported or run here: it pulls
in torchaudio).  The epoch-loop test (tests/test_trainer_cpu.py) and the GPU call-sequence test
(tests/test_dataset_gpu.py::test_main_py_call_sequence) execute the same sequence with real data."""
import importlib
import inspect
import json

import pytest


@pytest.fixture(scope="module")
def contract(golden_dir):
    return json.loads((golden_dir / "main_contract.json").read_text())


def test_every_imported_name_exists(contract):
    assert set(contract["imports"]) == {"config", "utils", "dataset", "trainer"}
    for module, names in contract["imports"].items():
        mod = importlib.import_module(module)
        for name in names:
            assert hasattr(mod, name), f"{module}.{name}"


def test_calls_bind_with_the_reference_keywords(contract):
    where = {n: m for m, names in contract["imports"].items() for n in names}
    for name, call in contract["calls"].items():
        if name == "DataLoader":
            continue                                       # torch's own
        fn = getattr(importlib.import_module(where[name]), name)
        sig = inspect.signature(fn.__init__ if inspect.isclass(fn) else fn)
        params = list(sig.parameters.values())
        if inspect.isclass(fn):
            params = params[1:]                            # self
        accepted = {p.name for p in params if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)}
        assert set(call["keywords"]) <= accepted, (name, set(call["keywords"]) - accepted)
        positional = [p for p in params if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        assert call["n_positional"] <= len(positional), name
        # nothing the reference leaves out may be required
        required = {p.name for p in params if p.default is p.empty and p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)}
        given = set(call["keywords"]) | {p.name for p in positional[:call["n_positional"]]}
        assert required <= given, (name, required - given)


def test_config_attributes_and_result_keys(contract, tmp_path):
    import config
    import trainer
    cfg = config.Config()
    for attr in contract["config_attributes"]:
        assert hasattr(cfg, attr), attr
    assert (cfg.CHECKPOINT_PATH / "best_model.pth").name == "best_model.pth"       # main.py:89: a Path
    assert contract["unpacked_results"] == {"load_files": 4, "setup_logging": 2, "train_model": 2}
    # the keys main.py reads are produced by the epoch loop / the tester (names checked in the source of truth: the
    # functions' own return statements, exercised end to end by tests/test_trainer_cpu.py)
    src_train = inspect.getsource(trainer.train_model)
    for key in contract["dict_keys"]["history"]:
        assert f'"{key}"' in src_train, key
    src_test = inspect.getsource(trainer.test_model)
    for key in contract["dict_keys"]["test_results"]:
        needle = 'f"class_{config.LOSS_TYPE}"' if key == "class_{}" else f'"{key}"'
        assert needle in src_test, key


def test_tuned_gemm_table_is_well_formed_and_inert_on_the_cpu():
    """seld_tuned.py: the shipped table carries the library versions it was tuned with and only GEMM rows; on a CPU
    device nothing is switched on."""
    import torch
    import seld_tuned
    rows = [line.split(",") for line in seld_tuned.TABLE.read_text().splitlines() if line]
    validators = {r[1] for r in rows if r[0] == "Validator"}
    assert {"PT_VERSION", "HIPBLASLT_VERSION", "ROCBLAS_VERSION", "GCN_ARCH_NAME"} <= validators
    assert any(r[1] == "GCN_ARCH_NAME" and r[2].startswith("gfx950") for r in rows)
    entries = [r for r in rows if r[0] != "Validator"]
    assert entries and all(r[0].startswith("Gemm") and len(r) == 4 for r in entries)
    before = dict(seld_tuned.state)
    assert seld_tuned.enable(torch.device("cpu")) == before
