# Round-end regression on the GPU box: smoke(), the GPU parity suite, the default bench line.
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke ok"
python -m pytest tests -q -m gpu 2>&1 | tail -1
python bench.py 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],2), 'ms', round(d['value']/1e6,1), 'M edges/s', 'frac', d['roofline']['frac'], 'cpu', round(d['cpu_baseline']['value']))"
