"""tests/golden/expected/ holds the oracle's output for a fixed list of command lines over the reference's fixture BAMs
(written by tests/golden/make_expected.py; NOT reference outputs).  CPU: the oracle still produces exactly these files
(no silent drift between rounds).  GPU: so does the product."""
import importlib.util
import pathlib

import pytest

import methyldackel_amd as mdk

HERE = pathlib.Path(__file__).resolve().parent / "golden"
spec = importlib.util.spec_from_file_location("make_expected", HERE / "make_expected.py")
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)


def expected(name):
    return {p.name[len(name) + 1:]: p.read_bytes() for p in (HERE / "expected").iterdir() if p.name.startswith(name + ".")}


@pytest.mark.parametrize("name", list(mk.COMMANDS))
def test_oracle_has_not_drifted(name):
    assert mk.run(name) == expected(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mk.COMMANDS))
def test_product_writes_the_same_files(name):
    assert mk.run(name, tool=mdk.CLI) == expected(name)
