python -m pytest tests/test_hip_train.py -x -q -m gpu 2>&1 | tail -1
python tools/determinism_check.py train 2>&1 | tail -1
for i in 1 2; do python bench.py --mode train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['train_step']['ms_per_iter'])"; done
