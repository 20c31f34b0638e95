#!/bin/bash
# Settle byte-level parity with a REAL MethylDackel 0.6.1 binary in one command:
#   METHYLDACKEL_BIN=/path/to/MethylDackel tests/golden/with_reference.sh
# Runs every command line of tests/golden/make_expected.py (the reference's 15 test vectors' inputs and more of the option
# surface, incl. vector t8's `--nOT 50,50,40,40`) through that binary on copies of the fixtures and diffs each output file
# against tests/golden/expected/ (this repo's oracle, which the GPU path is byte-compared with).  Exit 0 = identical.
# The same check runs under pytest (tests/test_reference_binary.py) and is skipped when the variable is unset.
set -e
cd "$(dirname "$0")/../.."
if [ -z "$METHYLDACKEL_BIN" ] || [ ! -x "$METHYLDACKEL_BIN" ]; then echo "set METHYLDACKEL_BIN to a MethylDackel 0.6.1 executable" >&2; exit 2; fi
"$METHYLDACKEL_BIN" --version || true
exec python -m pytest tests/test_reference_binary.py -q -rA
