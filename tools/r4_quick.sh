Q="--no-extra-legs --no-cpu-baseline --no-tcp-baseline --no-rtt --no-small-ring --steps 20 --warmup 3"
for w in staged direct; do python bench.py --wire $w $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w', d['value'], {k:v['us_per_launch'] for k,v in r['schedule_kernels'].items()}, 'frac', r['frac'], 'step', r['step_level']['frac'])"; done
