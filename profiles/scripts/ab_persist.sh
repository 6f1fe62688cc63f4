for f in 0 0; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('flags $f', d['value'], d['stage_ms'])"
done
AX_WHISPER_DECODE=graph timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('graph', d['value'], d['stage_ms'])"
