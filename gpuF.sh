cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py --steps 300 --warmup 50 2>&1 | tail -1 | cut -c1-400
