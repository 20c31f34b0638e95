"""Byte-level pin against a real reference binary, when one is available: set METHYLDACKEL_BIN to a MethylDackel 0.6.1 executable (the
reference cannot be built in this image: no htslib, no libBigWig).  Every command line of tests/golden/make_expected.py is run through it
on copies of the fixtures (the reference writes .fai/.bai files next to its inputs) and each file it writes must equal
tests/golden/expected/ -- the oracle's output, which the product is byte-compared with on the GPU.  Skipped without the variable.
tests/golden/with_reference.sh is the one-command form."""
import importlib.util
import os
import pathlib
import shutil
import subprocess
import tempfile

import pytest

HERE = pathlib.Path(__file__).resolve().parent / "golden"
spec = importlib.util.spec_from_file_location("make_expected", HERE / "make_expected.py")
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)
BIN = os.environ.get("METHYLDACKEL_BIN")
pytestmark = pytest.mark.skipif(not BIN, reason="METHYLDACKEL_BIN is not set (no reference binary in this image)")


def run_reference(name):
    cmd, args = mk.COMMANDS[name]
    with tempfile.TemporaryDirectory() as d:
        d = pathlib.Path(d); (d / "in").mkdir(); (d / "run").mkdir()
        for f in HERE.iterdir():
            if f.suffix in (".fa", ".bam", ".bai"):
                shutil.copy(f, d / "in" / f.name)
        a = [str(d / "in" / x) if x in mk.FIXTURES else x for x in args] + (["-o", "out"] if cmd == "extract" else [])
        r = subprocess.run([BIN, cmd] + a, cwd=d / "run", capture_output=True)
        assert r.returncode == 0, (name, r.stderr.decode()[-800:])
        out = {p.name: p.read_bytes() for p in sorted((d / "run").iterdir())}
        out["stdout"] = r.stdout
        if cmd == "mbias":
            out["suggestion"] = b"".join(l + b"\n" for l in r.stderr.splitlines() if l.startswith(b"Suggested"))
        return out


@pytest.mark.parametrize("name", list(mk.COMMANDS))
def test_reference_binary_writes_the_expected_files(name):
    want = {p.name[len(name) + 1:]: p.read_bytes() for p in (HERE / "expected").iterdir() if p.name.startswith(name + ".")}
    got = run_reference(name)
    assert sorted(got) == sorted(want), (sorted(got), sorted(want))
    for k in want:
        assert got[k] == want[k], f"{name}.{k} differs from the reference binary's output"
