"""Fold rocprofv3 counter_collection CSVs: per kernel (short name) x counter -> mean value per launch."""
import csv, glob, os, re, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r'([A-Za-z_][A-Za-z0-9_]*)(<[^(]*>)?\(', row['Kernel_Name'])
            k = (m.group(1) + (m.group(2) or '')) if m else row['Kernel_Name']
            if 'conv' not in k: continue
            k += ' grid=' + row.get('Grid_Size', '?')
            e = agg[k][row['Counter_Name']]; e[0] += 1; e[1] += float(row['Counter_Value'])
for k, cs in sorted(agg.items()):
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f'   {c:32s} {v / n:16.1f}  (n={n})')
