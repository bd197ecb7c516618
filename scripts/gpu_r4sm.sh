#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
time (timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -3)
time (timeout 900 python bench.py 2>/dev/null | cut -c1-200)
