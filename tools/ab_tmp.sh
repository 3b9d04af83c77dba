b() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); s=d['roofline']['stage_ms']; print(d['value'], s['decode_ms'], s['total_ms'])"; }
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
echo new; b; b
cp rtlsdr_ft8d_amd/csrc/decode.hip /tmp/new.hip; cp tools/decode_prev.txt rtlsdr_ft8d_amd/csrc/decode.hip; make -s -C rtlsdr_ft8d_amd/csrc -j16 >/dev/null 2>&1; echo prev; b; b
cp /tmp/new.hip rtlsdr_ft8d_amd/csrc/decode.hip; make -s -C rtlsdr_ft8d_amd/csrc -j16 >/dev/null 2>&1; echo new; b
