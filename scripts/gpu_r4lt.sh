ROOT=${GRAFT_REPO_ROOT:-$PWD}; export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd; cd $ROOT; timeout 100 python scripts/loc_trace.py 2>&1 | grep -v amdgpu
