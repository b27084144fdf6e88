cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_56; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json.txt 2> $O/bench.err; python - <<'PY'
import json
for l in open('gpurun_out/r03_56/bench.json.txt'):
    if l.startswith('{'):
        d=json.loads(l); m=d['more']
        print('value',d['value'],'frac',d['roofline']['frac'],'floor',d['roofline'].get('frac_of_box_floor'))
        print('pairs32',m['pairs_u32']['value'],'pairs64',m['pairs_u64']['value'],'keys64',m['keys64']['value'])
        print('entropy keys',[round(e['value'],1) for e in m['entropy_sweep']['keys']],'u64',[round(e['value'],1) for e in m['entropy_sweep']['pairs_u64']])
        print('size keys',[(e['log2_keys'],e['GKeys_per_s']) for e in m['size_sweep']['keys']])
        print('size pairs',[(e['log2_keys'],e['GKeys_per_s']) for e in m['size_sweep']['pairs_u32']])
PY
