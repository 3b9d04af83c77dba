b() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); s=d['roofline']['stage_ms']; print(d['value'], s['decode_ms'], s['total_ms'])"; }
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
b; b; b
