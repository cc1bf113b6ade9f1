for sk in 0.12 0.18 0.25 0.32 0.4; do
  CVO_HIP_SKIN=$sk timeout -k 10 150 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('skin=$sk', round(d['value'],1), 'align/s')"
done
