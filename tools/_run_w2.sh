set -o pipefail
for cfg in "10 50" "50 200" "10 50"; do set -- $cfg; python bench.py --no-cpu-baseline --warmup $1 --steps $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('warmup',d['warmup'],'steps',d['steps'],'ms',d['ms_per_step'],'wait',d['host_wait_ms_per_step'],'step_ms',d['step_ms'])"; done
