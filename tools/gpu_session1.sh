#!/bin/bash
# round-2 first GPU session: tests, default bench, copy/kernel overlap trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s1
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log
(timeout 600 python bench.py 2> $O/bench.err | grep "^{") > $O/bench.json
tail -5 $O/bench.err > $O/bench.err.tail; rm -f $O/bench.err
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/p_s1 -o b -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt --no-encoder --traffic off) > $O/trace.log 2>&1
find /tmp/p_s1 -name "*stats*.csv" -exec cp {} $O/ \;
python - <<PY > $O/overlap.txt 2>&1
import csv, glob
ks=[]; cs=[]
for p in glob.glob('/tmp/p_s1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
for p in glob.glob('/tmp/p_s1/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        cs.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Direction', r.get('Operation','?')), r))
ks.sort(); cs.sort()
t0 = min([k[0] for k in ks]+[c[0] for c in cs])
big=[k for k in ks if 'rd_lstm' in k[2]]
print('lstm launches', len(big))
for k in big[-10:]:
    print('K %10.3f %10.3f ms  dur %.3f' % ((k[0]-t0)/1e6, (k[1]-t0)/1e6, (k[1]-k[0])/1e6))
print('copies', len(cs))
for c in cs[-24:]:
    print('C %10.3f %10.3f ms dur %.3f %s' % ((c[0]-t0)/1e6, (c[1]-t0)/1e6, (c[1]-c[0])/1e6, c[2]))
names={}
for k in ks:
    names.setdefault(k[2],[0,0]); names[k[2]][0]+=1; names[k[2]][1]+=(k[1]-k[0])/1e6
for n,v in sorted(names.items(), key=lambda x:-x[1][1])[:15]:
    print('%-62s %6d %10.3f ms' % (n, v[0], v[1]))
PY
grep -h "^{" $O/trace.log | head -1 > $O/bench_under_trace.json
tail -3 $O/trace.log > $O/trace.tail; rm -f $O/trace.log
ls -la $O
