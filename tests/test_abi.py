"""CPU-only: both shared libraries load and export every symbol their headers declare (no compute calls)."""
import ctypes as C
import re
import subprocess

import methyldackel_amd as mdk
from conftest import REPO


def declared(header):
    txt = open(REPO / "include" / header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:md_|mdk_|extract_main|mbias_main|perRead_main|mergeContext_main)\w*)\s*\(", txt)))


def exported(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", str(lib)], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def test_hip_library_exports_header():
    want = declared("mdk_hip.h")
    assert set(want) == set(mdk.HIP_SYMBOLS), (sorted(set(want) ^ set(mdk.HIP_SYMBOLS)))
    have = exported(mdk.LIB_HIP)
    assert not [s for s in want if s not in have]
    L = mdk.lib_hip()
    for s in want:
        getattr(L, s)


def test_extract_library_exports_header():
    want = declared("mdk_extract.h")
    assert set(want) == set(mdk.EXTRACT_SYMBOLS), (sorted(set(want) ^ set(mdk.EXTRACT_SYMBOLS)))
    have = exported(mdk.LIB_EXTRACT)
    assert not [s for s in want if s not in have]
    L = mdk.lib_extract()
    for s in want:
        getattr(L, s)


def test_struct_layouts_match_the_c_abi():
    assert C.sizeof(mdk.md_seg) == 32
    assert C.sizeof(mdk.md_site) == 16 and C.sizeof(mdk.md_site_var) == 8 and C.sizeof(mdk.md_tile_seg) == 8
    assert C.sizeof(mdk.md_dev_cfg) == 4 * (5 + 32 + 3)


def test_no_device_is_a_loud_error_not_a_fallback(tmp_path):
    """on a box without a GPU the product must refuse to run (this test is skipped where a GPU exists)"""
    L = mdk.lib_hip()
    if L.md_dev_count() > 0:
        return
    from conftest import GOLDEN
    r = mdk.run_cli([str(GOLDEN / "cg100.fa"), str(GOLDEN / "cg_aln.bam"), "-q", "2", "-o", str(tmp_path / "x")])
    assert r.returncode == (-20) & 0xFF and "no CPU path" in r.stderr
    assert not (tmp_path / "x_CpG.bedGraph").exists() or (tmp_path / "x_CpG.bedGraph").read_text().count("\n") <= 1


def test_product_does_not_link_the_oracle():
    for lib in (mdk.LIB_HIP, mdk.LIB_EXTRACT, mdk.CLI):
        out = subprocess.run(["ldd", str(lib)], capture_output=True, text=True).stdout
        assert "oracle" not in out
    src = subprocess.run(["grep", "-rl", "oracle/", str(REPO / "methyldackel_amd")], capture_output=True, text=True).stdout
    assert src.strip() == "", f"product sources must not reference oracle/: {src}"
