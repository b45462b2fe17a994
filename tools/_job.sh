cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gp_oracle.py -q -x 2>&1 | tail -5
