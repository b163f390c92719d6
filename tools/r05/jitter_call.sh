bash tools/r05/jitter.sh 2>&1 | tail -70; python -m pytest tests/test_sort_gpu.py -m gpu -q -k "two_valued" 2>&1 | tail -2; python bench.py > gpurun_out/r05_jitter/bench.json 2> gpurun_out/r05_jitter/bench.err; python3 -c "
import json; d=json.load(open('gpurun_out/r05_jitter/bench.json')); print(d['value'], d['key_value']['value'], d['setup'], d['roofline']['traffic'])"
