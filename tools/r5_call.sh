cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_math_gpu.py -m gpu -q 2>&1 | tail -8 | cut -c1-250
