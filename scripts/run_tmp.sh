cd $GRAFT_REPO_ROOT
timeout 400 python scripts/exp_two_ctx.py 2>&1 | tail -4
