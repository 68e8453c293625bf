cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -5
