"""The mappability rule at its rounding edges (reference extract.c:1138-1144,1188), against HAND-COMPUTED values -- not against a track the
same generator wrote ("a wrong rounding rule would pass").

  val  = (unsigned char)(char)((double)raw * 100 + 0.5), NaN -> 0          (raw is the bigWig's float, widened to double first)
  bit  = val >= (double)cutoff_float * 100.0                               (--mappabilityThreshold, default 0.01)

So 0.005f (= 0.00499999988...) gives 0.99999998... -> 0, not the 1 a float product (0.5f + 0.5) would give; 0.015f gives 1; 0.0149 gives 1;
0.0051 gives 1; 0.995f (= 0.99500000477) gives 100; NaN (uncovered) gives 0.  The product writes `val` per base into the BBM it re-encodes the bigWig as (-O), which is where
this test reads it back; the bits are then checked through the admission of reads placed on those bases (-M with --minMappableBases)."""
import struct
import subprocess
from fractions import Fraction

import numpy as np
import pytest

import methyldackel_amd as mdk
from bwwriter import write_bigwig

VALUES = [0.005, 0.0051, 0.0149, 0.015, 0.0151, 0.0249, 0.025, 0.5, 0.994, 0.995, 0.9951, 1.0, 0.0, 0.00499, 0.0050001]


def expected_val(v):
    d = float(np.float32(v))                       # what the file holds, widened to double
    got = int(d * 100 + 0.5)                       # (char) of a double in 0..101: truncation toward zero
    assert got == int(Fraction(d) * 100 + Fraction(1, 2)), v      # none of the test values sits within rounding error of a boundary: exact arithmetic agrees
    return got


def read_bbm(path):
    """-> {name: [val per base]} (BBM_Specification.md: version byte, u32 contigs; per contig u16 name length, name, 0, u32 length, run-length coded values)"""
    d = open(path, "rb").read(); assert d[0] == 1
    n = struct.unpack_from("<I", d, 1)[0]; o = 5; out = {}
    for _ in range(n):
        nl = struct.unpack_from("<H", d, o)[0]; o += 2
        name = d[o:o + nl].decode(); o += nl + 1
        ln = struct.unpack_from("<I", d, o)[0]; o += 4
        vals = []
        while len(vals) < ln:
            v = d[o]; o += 1; run = 1
            if v > 100:
                if v == 255:
                    run = struct.unpack_from("<H", d, o)[0]; o += 2
                else:
                    run = v - 99
                v = d[o]; o += 1
            vals += [v] * run
        out[name] = vals[:ln]
    return out


def test_hand_computed_expectations():
    # the arithmetic this file asserts, spelled out for the values that decide the rule
    assert expected_val(0.005) == 0 and expected_val(0.0051) == 1 and expected_val(0.0149) == 1 and expected_val(0.015) == 1 and expected_val(0.0151) == 2
    assert expected_val(0.994) == 99 and expected_val(0.995) == 100 and expected_val(1.0) == 100       # 0.995f is 0.99500000477: it rounds up, 0.005f (0.00499999989) down
    assert int(np.float32(np.float32(0.005) * np.float32(100)) + 0.5) == 1           # the float-product rule would say 1 here: that is the wrong rule


def test_bigwig_values_at_the_rounding_edges(tmp_path):
    runs = [(10 * i, 10 * i + 7, v) for i, v in enumerate(VALUES)]                    # 3 uncovered bases (NaN) after each run
    write_bigwig(tmp_path / "edge.bw", [("chrE", 10 * len(VALUES) + 5)], {0: runs})
    r = subprocess.run([str(mdk.CLI), "extract", "-M", str(tmp_path / "edge.bw"), "-N", str(tmp_path / "edge")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    vals = read_bbm(tmp_path / "edge.bbm")["chrE"]
    for i, v in enumerate(VALUES):
        assert vals[10 * i:10 * i + 7] == [expected_val(v)] * 7, (v, vals[10 * i:10 * i + 7])
        assert vals[10 * i + 7:10 * i + 10] == [0, 0, 0]                               # NaN -> 0
