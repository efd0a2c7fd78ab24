# scratch command file for ad-hoc gpurun experiments (kept empty in the tree)
cd $GRAFT_REPO_ROOT
