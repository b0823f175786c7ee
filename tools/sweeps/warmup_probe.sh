for i in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('w5 s20', d['ms_per_step'])"
python bench.py --gpus 1 --steps 20 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('w100 s20', d['ms_per_step'])"
python bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('w5 s200', d['ms_per_step'])"
done
