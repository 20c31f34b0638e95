"""mdk_bind_to_device_node (csrc/host/mdk_affinity.c): the command binds itself to the CPUs next to its GPU.  Exercised against a
fake sysfs tree (MDK_SYSFS_DRM), in a child process each time because the call changes the caller's affinity."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent

CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import methyldackel_amd as mdk
L = mdk.lib_extract()
rc = L.mdk_bind_to_device_node(int(sys.argv[1]))
print(rc, ",".join(map(str, sorted(os.sched_getaffinity(0)))))
""" % str(REPO)


def fake_tree(root, cards):
    for name, vendor, cpulist in cards:
        d = root / name / "device"
        d.mkdir(parents=True)
        if vendor is not None:
            (d / "vendor").write_text(vendor + "\n")
        if cpulist is not None:
            (d / "local_cpulist").write_text(cpulist + "\n")


def run(root, index, **env):
    r = subprocess.run([sys.executable, "-c", CHILD, str(index)], capture_output=True, text=True, env=dict(os.environ, MDK_SYSFS_DRM=str(root), **env))
    assert r.returncode == 0, r.stderr
    rc, cpus = r.stdout.split()
    return int(rc), [int(c) for c in cpus.split(",")]


def test_binds_to_the_gpus_local_cpus(tmp_path):
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 4:
        pytest.skip("needs at least 4 CPUs")
    half = len(allowed) // 2
    lo, hi = allowed[:half], allowed[half:]
    as_list = lambda cs: ",".join(map(str, cs))
    fake_tree(tmp_path, [("card0", "0x1002", as_list(lo)), ("card0-DP-1", None, None), ("card1", "0x1a03", as_list(allowed)),
                         ("card2", "0x1002", f"{hi[0]}-{hi[-1]}" if hi == list(range(hi[0], hi[-1] + 1)) else as_list(hi)), ("renderD128", "0x1002", as_list(lo))])
    assert run(tmp_path, 0) == (len(lo), lo)                     # first AMD GPU in PCI (here: path) order
    assert run(tmp_path, 1) == (len(hi), hi)                     # the VGA card of another vendor is not counted
    assert run(tmp_path, 2) == (0, allowed)                      # no such GPU: nothing changes
    assert run(tmp_path, 0, MDK_NO_BIND="1") == (0, allowed)
    assert run(tmp_path, 0, ROCR_VISIBLE_DEVICES="1") == (0, allowed)      # a renumbered device list: the index says nothing about sysfs order
    assert run(tmp_path, 0, HIP_VISIBLE_DEVICES="") == (len(lo), lo)       # (an empty variable is no restriction)


@pytest.mark.parametrize("cpulist", ["", "0-3,x", "3-1", "-2", "0", "99999"])
def test_leaves_the_process_alone_when_the_list_is_unusable(tmp_path, cpulist):
    allowed = sorted(os.sched_getaffinity(0))
    fake_tree(tmp_path, [("card0", "0x1002", cpulist)])
    assert run(tmp_path, 0) == (0, allowed)                      # malformed, a single CPU (< half), or outside the allowed set


def test_one_numa_node_is_a_no_op(tmp_path):
    allowed = sorted(os.sched_getaffinity(0))
    fake_tree(tmp_path, [("card0", "0x1002", ",".join(map(str, allowed)))])
    assert run(tmp_path, 0) == (0, allowed)


def test_missing_sysfs_is_a_no_op(tmp_path):
    assert run(tmp_path / "nothing_here", 0) == (0, sorted(os.sched_getaffinity(0)))
