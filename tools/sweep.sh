#!/bin/bash
# usage: sweep.sh N "nb:outer:tail nb:outer:tail ..."
N=$1; shift
for cfg in $@; do
  IFS=: read nb outer tail rsv bw d2 <<< "$cfg"; [ -z "$rsv" ] && rsv=-1; [ -z "$bw" ] && bw=-1; [ -z "$d2" ] && d2=-1
  r=$(python bench.py --n $N --nb $nb --outer $outer --tail $tail --reserve $rsv --depth2 $d2 --steps 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('%.2f TF  %.1f ms  kernel %.2f TF x%d avg %.2f ms  residual %.2e' % (d['value'], d['ms_per_step'], r.get('achieved',0), r.get('launches',0), r.get('avg_launch_ms',0), d['config'].get('residual',-1)))")
  echo "N=$N nb=$nb outer=$outer tail=$tail reserve=$rsv depth2=$d2 : $r"
done
