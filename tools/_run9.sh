set -eo pipefail
mkdir -p gpurun_out/w9
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "lrn_pool_fwd or pool_lrn" > gpurun_out/w9/test.log 2>&1 || { tail -40 gpurun_out/w9/test.log; exit 1; }
tail -1 gpurun_out/w9/test.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/w9/test_gpu.log 2>&1 || { tail -30 gpurun_out/w9/test_gpu.log; exit 1; }
tail -1 gpurun_out/w9/test_gpu.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/w9/bench.json 2> gpurun_out/w9/bench.err
python -c "
import json; r=json.load(open('gpurun_out/w9/bench.json')); print(r['value'], r['ms_per_step'], r['check'])"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/w9/prof -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/w9/prof_bench.json 2> gpurun_out/w9/prof_bench.err
python3 - <<'P'
import sqlite3,glob
c=sqlite3.connect(glob.glob('gpurun_out/w9/prof/*_results.db')[0])
for n,k,a in c.execute("select name,count(*),avg(duration) from kernels where name like '%lrn%' or name like '%pool%' group by name"):
    print(n[:60],k,round(a/1e3,1),'us')
P
