# Round-2 evidence of the bf16 conv path (BASELINE config 5).  Run on the GPU box from the repo root: bash tools/profile_bf16.sh
#   bench lines at 16- and 32-frame clips, kernel trace of the 16-frame step, PMC probe of the conv kernels (matrix-pipe busy,
#   VALU : MFMA instruction ratio, LDS bank conflicts)  ->  gpurun_out/bf16/
set -eo pipefail
out=gpurun_out/bf16
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --conv-math bf16 --no-cpu-baseline --no-side > $out/bench_fpc16.json 2> $out/bench_fpc16.err
python3 bench.py --conv-math bf16 --fpc 32 --no-cpu-baseline --no-side > $out/bench_fpc32.json 2> $out/bench_fpc32.err
rocprofv3 --kernel-trace --stats -d $out/prof -o b16 -- python3 bench.py --conv-math bf16 --no-cpu-baseline --no-side --steps 8 --warmup 2 > /dev/null 2> $out/prof.err
python3 - <<'P' > $out/kernel_stats.csv
import sqlite3,glob
c=sqlite3.connect(sorted(glob.glob('gpurun_out/bf16/prof/*.db'))[-1])
print("name,calls,total_us,avg_us,pct")
for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print('"%s",%d,%.1f,%.2f,%.2f' % (r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4]))
P
C8_OUT=bf16/pmc bash tools/c8_pmc_probe.sh > $out/conv_c8_pmc.txt 2>&1
tail -20 $out/conv_c8_pmc.txt
