cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_43; mkdir -p $O
( time timeout 900 python bench.py > $O/bench.json.txt 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time
python - <<'PY'
import json
for l in open('gpurun_out/r03_43/bench.json.txt'):
    if l.startswith('{'):
        d=json.loads(l); m=d['more']
        print('value',d['value'],'frac',d['roofline']['frac'],'floor',d['roofline'].get('frac_of_box_floor'))
        print('pairs32',m['pairs_u32']['value'],'pairs64',m['pairs_u64']['value'],'keys64',m['keys64']['value'])
        for k in ('keys','pairs_u32','pairs_u64'):
            print('entropy',k,[round(e['value'],1) for e in m['entropy_sweep'][k]])
        print('size keys',[(e['log2_keys'],e['GKeys_per_s']) for e in m['size_sweep']['keys'][8:]])
        print('size pairs',[(e['log2_keys'],e['GKeys_per_s']) for e in m['size_sweep']['pairs_u32'][8:]])
PY
timeout 600 python -m pytest tests/test_gpu_fault.py tests/test_gpu_midpath.py tests/test_gpu_tools.py -m gpu -q > $O/pytest_part.txt 2>&1; tail -1 $O/pytest_part.txt
