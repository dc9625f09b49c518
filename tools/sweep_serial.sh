#!/bin/bash
N=$1; shift
for sm in "$@"; do
  r=$(python bench.py --n $N --serial-m $sm --steps 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('%.2f TF  %.1f ms  kernel %.2f TF x%d  residual %.2e' % (d['value'], d['ms_per_step'], r.get('achieved',0), r.get('launches',0), d['config'].get('residual',-1)))")
  echo "N=$N serial_m=$sm : $r"
done
