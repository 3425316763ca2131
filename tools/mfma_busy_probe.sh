# Effective clock (GRBM_GUI_ACTIVE / 8 / duration) and matrix-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles)
# of single conv kernels at the benchmark size.  Run on the GPU box from the repo root: bash tools/mfma_busy_probe.sh
set -eo pipefail
out=gpurun_out/${MFMA_OUT:-mfma_busy}
mkdir -p $out
export TMPDIR=/tmp
for k in "conv3 fwd" "conv3 wgrad" "conv2 dgrad" "conv2 wgrad" "conv4 fwd" "conv1 fwd"; do
  set -- $k
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $out/$1_$2 -o runc -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py $1 $2 1024 3 > $out/$1_$2.log 2>&1
done
python3 - <<'P'
import sqlite3,glob
for d in sorted(glob.glob('gpurun_out/'+__import__("os").environ.get("MFMA_OUT","mfma_busy")+'/*/')):
    c=sqlite3.connect(glob.glob(d+'*_results.db')[0])
    q="select kernel_name, counter_name, avg(value), avg(duration), count(*) from counters_collection where (kernel_name like '%mfma_contract%' or kernel_name like '%wgrad_dma%' or kernel_name like '%conv_dma%') group by kernel_name, counter_name"
    res={}
    for n,cn,v,dur,k in c.execute(q):
        res.setdefault(n[5:50].split('(')[0],{})[cn]=v; res[n[5:50].split('(')[0]]['dur']=dur
    for n,r in res.items():
        if 'GRBM_GUI_ACTIVE' not in r: continue
        clk=r['GRBM_GUI_ACTIVE']/8/ r['dur']  # cycles per ns = GHz
        cyc=r['GRBM_GUI_ACTIVE']/8
        print('%-12s %-46s dur %.3f ms  clock %.3f GHz  matrix pipe busy %.1f %% of cycles   wave cycles: waiting to issue %.0f %%, waitcnt/barrier %.0f %%'
              % (d.split('/')[-2], n, r['dur']/1e6, clk, 100*r['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc,
                 100*r['SQ_WAIT_INST_ANY']/r['SQ_WAVE_CYCLES'], 100*r['SQ_WAIT_ANY']/r['SQ_WAVE_CYCLES']))
P
