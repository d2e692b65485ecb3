cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pb -o bn -- python /root/repo/tools/bn_bench.py > /dev/null 2>&1
f=$(find /tmp/pb -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'bn_' in n:
        agg[(n.split('(')[1][-30:] if False else n[:60], r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items()):
    v.sort()
    print('%-62s grid %-8s n=%4d  median %.1f us  min %.1f' % (k[0], k[1], len(v), v[len(v) // 2], v[0]))
PY
