#!/bin/bash
# full GPU validation: the whole -m gpu suite (log written directly: no pipe), smoke, default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
tag=${1:-full}
timeout -k 10 1000 python -m pytest tests -m gpu -q --tb=short > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/${tag}_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 500 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
