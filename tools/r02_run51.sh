set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
grep -E "passed|failed|rror" $O/pytest.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py > $O/bench.txt 2> $O/bench.err; tail -c 200 $O/bench.err; cut -c1-330 $O/bench.txt
