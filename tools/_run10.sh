cd $GRAFT_REPO_ROOT
python tools/diag_seed_scan.py 32 103 104 105 106 2>&1 | grep "^K"
python tools/diag_seed_scan.py 16 87 88 2>&1 | grep "^K"
