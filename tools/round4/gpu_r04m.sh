#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04m_pytest.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/r04m_pytest.log
timeout 900 python tools/round4/r04m.py > gpurun_out/r04m_stdout.txt 2> gpurun_out/r04m_stderr.txt; echo rc=$?
grep "^==" gpurun_out/r04m_e2e.txt | cut -c1-400; tail -3 gpurun_out/r04m_stderr.txt
