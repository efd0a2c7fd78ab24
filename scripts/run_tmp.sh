cd $GRAFT_REPO_ROOT
timeout 300 python scripts/cnn_variants.py 2>&1 | tail -12
