"""The oracle's `-@ N` mode -- chunk-parallel workers claiming chunks under a mutex and flushing in bin order, as the
reference's extractCalls threads do (extract.c:325-350,514-535,1479-1486) -- is the all-core CPU baseline of bench.py.
Its output must not depend on N."""
import subprocess

import pytest

from conftest import GOLDEN, ORACLE, synth
from t8_differential import NAMES, run


def _run(args, cwd, threads):
    cwd.mkdir()
    r = subprocess.run([str(ORACLE), "extract"] + [str(a) for a in args] + ["-@", str(threads), "-o", "out"], cwd=cwd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return {p.name: p.read_bytes() for p in sorted(cwd.iterdir())}, r.stdout


@pytest.mark.parametrize("extra", [[], ["--CHG", "--CHH", "--chunkSize", "7000"], ["--mergeContext", "--CHG", "--chunkSize", "2500"],
                                   ["--minOppositeDepth", "2", "--maxVariantFrac", "0.2", "--chunkSize", "9000"], ["--cytosine_report", "--chunkSize", "30000"]],
                         ids=["default", "allctx", "merge", "variant", "cytosine_report"])
def test_thread_count_does_not_change_output(tmp_path, extra):
    synth(tmp_path / "s", "-L", "90000,30000,500", "-c", "20", "-s", "21", "--extras")
    args = [tmp_path / "s.fa", tmp_path / "s.bam"] + extra
    one, out1 = _run(args, tmp_path / "t1", 1)
    for n in (3, 8):
        many, outn = _run(args, tmp_path / f"t{n}", n)
        assert many == one and outn == out1
    assert sum(v.count(b"\n") for v in one.values()) > 1000


def test_threads_on_reference_fixture(tmp_path):
    one, _ = _run([GOLDEN / "cg100.fa", GOLDEN / "cg_aln.bam", "-q", "2", "--chunkSize", "10"], tmp_path / "a", 1)
    four, _ = _run([GOLDEN / "cg100.fa", GOLDEN / "cg_aln.bam", "-q", "2", "--chunkSize", "10"], tmp_path / "b", 4)
    assert one == four and one["out_CpG.bedGraph"].count(b"\n") == 49


def test_t8_differential_search_result():
    """Reference vector t8 (tests/test.py:82-88, 12 lines) against single-rule deviations of the oracle: the only deviation
    that reproduces it is a right-hand absolute trim one base short of what common.c:198-204 does; every deviation of the
    overlap rule leaves t8 at 11 lines (a trimmed base has quality 0, and no overlap rule can raise a quality from 0)."""
    hits = []
    for pt in range(len(NAMES)):
        res = run(pt)
        if res["t8"][1]["_CpG.bedGraph"] == 12 and all(ok for k, (ok, _) in res.items() if k != "t8"):
            hits.append(pt)
    assert hits == [1]
