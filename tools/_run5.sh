set -eo pipefail
out=gpurun_out/w5
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o runc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.json 2> $out/pmc_fetch.err
python3 - <<'P'
import sqlite3,glob
c=sqlite3.connect(glob.glob('gpurun_out/w5/pmc_fetch/*_results.db')[0])
q="select kernel_name, grid_size, avg(value), avg(duration), count(*) from counters_collection where counter_name='FETCH_SIZE' and (kernel_name like '%ConvGather%' or kernel_name like '%wgrad%') group by kernel_name, grid_size order by min(start)"
for n,g,v,d,k in c.execute(q):
    print(n[5:60], g, k, round(v*2048/1e6,1),'MB', round(d/1e6,3),'ms')
P
