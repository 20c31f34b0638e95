"""The mate-overlap quality rule the kernels evaluate (methyldackel_amd/csrc/mdk_overlap_rule.h): the literal form of
cust_tweak_overlap_quality (overlaps.c:90-109) and its three-select form must be the same function, checked here over every
input (own-is-later x 16 x 16 bases x 256 x 256 qualities), and `boost` must be the C expression the reference evaluates."""
import subprocess
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent

HARNESS = r"""
#include <cstdio>
#include <cstdint>
#include "mdk_overlap_rule.h"
int main() {
    long bad = 0, n = 0;
    for(int q = 0; q < 256; q++) if(boost(q) != (int)(uint8_t)(q + 0.2 * q)) bad++;          /* overlaps.c:103,106 */
    for(int s = 0; s < 2; s++) for(int bo = 0; bo < 16; bo++) for(int bm = 0; bm < 16; bm++) for(int x = 0; x < 256; x++) for(int y = 0; y < 256; y++) {
        n++; if(resolve_overlap(s, bo, x, bm, y) != resolve_own(s, bo, x, bm, y)) bad++;
    }
    printf("%ld %ld\n", n, bad);
    return bad != 0;
}
"""


def test_select_form_equals_the_literal_rule(tmp_path):
    (tmp_path / "h.cpp").write_text(HARNESS)
    subprocess.run(["g++", "-O2", "-I", str(REPO / "methyldackel_amd/csrc"), "-o", str(tmp_path / "h"), str(tmp_path / "h.cpp")], check=True)
    r = subprocess.run([str(tmp_path / "h")], capture_output=True, text=True)
    n, bad = map(int, r.stdout.split())
    assert r.returncode == 0 and bad == 0 and n == 2 * 16 * 16 * 256 * 256, r.stdout
