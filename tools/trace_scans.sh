# per-call durations of the rocprim scan kernels in one bench step, with the kernel before and after each (run through gpurun)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/prof_tr/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
import re
def short(n):
    m = re.findall(r'(k_[a-z_0-9]+|lookback_scan_kernel|partition_kernel|onesweep|histogram|fillBuffer|copyBuffer|init_lookback|transform_kernel|[a-z_]+_kernel)', n)
    return ('/'.join(dict.fromkeys(m)) or n[:50])[:60]
agg = collections.OrderedDict()
for i, r in enumerate(rows):
    n = r['Kernel_Name']
    if 'rocprim' in n:
        prev = short(rows[i-1]['Kernel_Name']) if i else ''
        nxt = short(rows[i+1]['Kernel_Name']) if i + 1 < len(rows) else ''
        key = (short(n) + ' <- ' + prev[-40:], nxt[-40:], r.get('Grid_Size_X', r.get('Grid_Size','')), r.get('Stream_Id',''))
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%4d calls %9.1f us total %8.1f us avg | after %s | before %s | grid %s stream %s" % (c, t, t / c, k[0], k[1], k[2], k[3]))
PY
