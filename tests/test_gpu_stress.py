"""Short commands back to back, every command x both preparation modes, from a process that itself holds the device: none may
end with anything but the expected output.  (Round 4's driver run lost `MethylDackel mbias` with MDK_HOST_PREP=1 on a 60 kb input to a
GPU exception once in 281 tests; tools/round5/stress.py is the long form of this loop, profiles/r05a_stress.json its log.)"""
import hashlib
import os

import pytest

import methyldackel_amd as mdk

pytestmark = pytest.mark.gpu

RUNS = int(os.environ.get("MDK_STRESS_RUNS", "25"))


def outputs(d):
    h = hashlib.sha256()
    for f in sorted(d.iterdir()):
        if f.is_file():
            h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def held_device(small_synth, tmp_path_factory):
    """the pytest process keeps a handle of its own open while the commands run, as after any in-process C-ABI test"""
    plan = mdk.Plan([str(small_synth / "pe.fa"), str(small_synth / "pe.bam"), "-o", str(tmp_path_factory.mktemp("held") / "x")])
    dev = mdk.Device(plan.dev_cfg())
    yield dev
    dev.close(); plan.close()


@pytest.mark.parametrize("env", [{"MDK_HOST_PREP": "1"}, {}], ids=["host-prep", "device-prep"])
@pytest.mark.parametrize("command", ["mbias", "extract", "perRead"])
def test_repeated_short_command(tmp_path, small_synth, held_device, command, env):
    fa, bam = str(small_synth / "pe.fa"), str(small_synth / "pe.bam")
    args = {"mbias": [fa, bam, "--CHG", "--chunkSize", "7000", "--noSVG", "--txt"],
            "extract": [fa, bam, "--CHG", "--chunkSize", "7000", "-o", "x"],
            "perRead": [fa, bam, "--chunkSize", "7000", "-o", "pr.txt"]}[command]
    want = None
    for i in range(RUNS):
        d = tmp_path / f"r{i}"; d.mkdir()
        r = mdk.run_cli(args, cwd=d, env=env, command=command, timeout=120)
        assert r.returncode == 0, (i, r.returncode, r.stderr[-2000:])
        got = (r.stdout, outputs(d))
        if want is None:
            want = got
        assert got == want, f"run {i} differs from run 0"
