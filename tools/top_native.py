import json,sys
d=json.load(open(sys.argv[1]))
for c in d['cells']:
    nat=[r for r in c['top5'] if r['native']]
    print(c['shape'],c['m'],'dropped',len(c['dropped']))
    for d_ in c['dropped'][:5]: print('   DROPPED',d_)
import csv
rows=list(csv.DictReader(open(sys.argv[1].replace('.json','.csv'))))
from collections import defaultdict
by=defaultdict(list)
for r in rows:
    sid=int(r['solution'],16)
    if (sid>>48)&0xF in (9,13): by[(r['shape'],r['m'],(sid>>32)&7)].append((float(r['us_median']),r['solution'],r['desc'][9:120]))
for k,v in by.items():
    v.sort()
    print(k)
    for t in v[:6]: print('   %.2f %s %s'%t)
