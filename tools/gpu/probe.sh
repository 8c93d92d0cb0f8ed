cd $GRAFT_REPO_ROOT
tools/_probe/e32 48 128 128 256 256 3 3 256 | tail -1
tools/_probe/e544 48 128 128 256 256 3 3 256 | tail -1
tools/_probe/e544 48 128 128 256 256 3 3 256 | head -1
tools/_probe/e32 48 128 128 256 256 3 3 256 | head -1
