python tools/bench_configs.py hetero_20_8_10 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('@@CFG@@'):
        d=json.loads(ln[7:]); print('alone:', d['hetero_20_8_10']['ms'])"
python tools/bench_configs.py sweep_20_8_50 hetero_20_8_10 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('@@CFG@@'):
        d=json.loads(ln[7:]); print('after sweep_20_8_50:', d['hetero_20_8_10']['ms'])"
python tools/bench_configs.py config3 hetero_20_8_10 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('@@CFG@@'):
        d=json.loads(ln[7:]); print('after config3:', d['hetero_20_8_10']['ms'])"
python tools/bench_configs.py 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('@@CFG@@'):
        d=json.loads(ln[7:]); print('after all:', d['hetero_20_8_10']['ms'], d['tracking_12_8_30']['ms'])"
