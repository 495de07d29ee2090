cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests/test_masking_gpu.py -q --timeout 300 -x -k "ties" 2>&1 | grep -E "Error|error|assert|Mismatch|mask of|weight of|Max|x:|y:" | head -30
