cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_spatial.py -m gpu -q -x -k "reacher" 2>&1 | tail -8
