#!/bin/bash
# debug helper: reproduce tests/test_gpu_prep.py::test_names_with_many_records under rocgdb
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import sys, pathlib
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from test_gpu_prep import many_records_one_name
d=pathlib.Path('/tmp/many'); d.mkdir(exist_ok=True)
print(many_records_one_name(d,16))
PY
cd /tmp/many
MDK_HOST_PROFILE=1 /opt/rocm/bin/rocgdb -batch -ex run -ex bt -ex "info threads" --args ${GRAFT_REPO_ROOT}/methyldackel_amd/_build/MethylDackel extract m.fa m.bam -F 0 -q 0 --keepDupes -o out 2>&1 | tail -40
