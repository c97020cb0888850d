cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ab2.sh -n 60 k2w16 k2w16q4 2>&1 | tail -8
