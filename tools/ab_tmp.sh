b() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['stage_ms']['sync_ms'], d['roofline']['stage_ms']['total_ms'], d['config']['decoded_messages_per_frame'])"; }
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
b; b; b
