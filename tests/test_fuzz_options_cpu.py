"""CPU: random `extract` command lines through the oracle and through the command on the device stand-in (tools/round5/fuzz_options.py: option
parsing, chunk schedule, variant filter, context merging, the emitters' formats, messages and exit codes are the product's own code; the
counting is looked up from the oracle's dump): same exit code, same files byte for byte, same first line of the refusal.  Two fixed seeds
here; the tool takes any."""
import subprocess
import sys

import pytest

from conftest import REPO


@pytest.mark.parametrize("seed", [11, 12])
def test_random_command_lines_equal_the_oracle(tmp_path, seed):
    subprocess.run(["make", "-C", str(REPO), "tools/_build/libmdk_dev_standin.so", "tools/_build/mdk_synth"], check=True, capture_output=True)
    r = subprocess.run([sys.executable, str(REPO / "tools/round5/fuzz_options.py"), str(seed), "25", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "25 command lines" in r.stdout and " 0 differing" in r.stdout


def test_random_runs_of_the_ranks_driver_equal_the_oracle(tmp_path):
    """the same for the one-process-per-GPU driver (csrc/host/mdk_ranks.c; tools/round5/fuzz_ranks.py): 2-4 ranks on this host, random chunk sizes
    and options, chunks dealt or claimed, with or without the index, chunks handed back to the host"""
    subprocess.run(["make", "-C", str(REPO), "tools/_build/libmdk_dev_standin.so", "tools/_build/mdk_synth"], check=True, capture_output=True)
    r = subprocess.run([sys.executable, str(REPO / "tools/round5/fuzz_ranks.py"), "13", "10", str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 differing" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


def test_random_command_lines_through_the_host_preparation_equal_the_oracles_counters(tmp_path):
    """the host preparation (the path of a chunk the device hands back, and of MDK_HOST_PREP=1) over a random data shape and random options
    that bear on the counting: its packed batches, evaluated by tests/batch_eval.py, against the oracle's per-column counters
    (tools/round5/fuzz_host_prep.py)"""
    subprocess.run(["make", "-C", str(REPO), "tools/_build/mdk_synth"], check=True, capture_output=True)
    r = subprocess.run([sys.executable, str(REPO / "tools/round5/fuzz_host_prep.py"), "14", "10", str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 differing" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
