# Matrix-pipe utilisation, wait split, VALU : MFMA instruction ratio and LDS bank conflicts of the packed-bf16 conv kernels at the
# benchmark size (tools/c8_probe.py).  Run on the GPU box from the repo root: bash tools/c8_pmc_probe.sh [layers]
set -eo pipefail
out=gpurun_out/${C8_OUT:-c8_pmc}
mkdir -p $out
export TMPDIR=/tmp
layers=${1:-conv2,conv3,conv4,conv5}
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $out/a -o runc -- python3 $GRAFT_REPO_ROOT/tools/c8_probe.py 1024 2 $layers > $out/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU -d $out/b -o runc -- python3 $GRAFT_REPO_ROOT/tools/c8_probe.py 1024 2 $layers > $out/b.log 2>&1
python3 - <<'P'
import sqlite3,glob,os
out='gpurun_out/'+os.environ.get("C8_OUT","c8_pmc")
res={}
for part in ('a','b'):
    c=sqlite3.connect(glob.glob(out+'/'+part+'/*_results.db')[0])
    q="select kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name, avg(value), avg(duration) from counters_collection where (kernel_name like '%conv_c8_kernel%' or kernel_name like '%wgrad_c8_kernel%') group by kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name"
    try:
        rows=list(c.execute(q))
    except Exception as e:
        q=q.replace("grid_size_x, grid_size_y, grid_size_z,","grid_size,").replace("grid_size_x, grid_size_y, grid_size_z","grid_size")
        rows=[(r[0],r[1],0,0)+tuple(r[2:]) for r in c.execute(q)]
    for n,gx,gy,gz,cn,v,dur in rows:
        key=(n.split('(')[0].replace('void ',''),gx,gy,gz)
        res.setdefault(key,{})[cn]=v
        res[key]['dur_'+part]=dur
for key,r in sorted(res.items()):
    if 'GRBM_GUI_ACTIVE' not in r: continue
    cyc=r['GRBM_GUI_ACTIVE']/8
    line='%-34s grid %-14s %.3f ms  matrix pipe busy %4.1f %%  waves: waiting to issue %2.0f %%, waitcnt/barrier %2.0f %%' % (
        key[0], 'x'.join(str(int(k)) for k in key[1:]), r['dur_a']/1e6, 100*r['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc,
        100*r['SQ_WAIT_INST_ANY']/r['SQ_WAVE_CYCLES'], 100*r['SQ_WAIT_ANY']/r['SQ_WAVE_CYCLES'])
    if 'SQ_INSTS_MFMA' in r:
        # SQ_INSTS_VALU counts the MFMAs too (they are vector-ALU instructions): "other" = everything that is not an MFMA, whole kernel
        line+='  SQ_INSTS_VALU/SQ_INSTS_MFMA %.2f (other VALU per MFMA %.2f)  LDS/MFMA %.2f  bank-conflict cycles %.1f %% of LDS cycles' % (
            r['SQ_INSTS_VALU']/max(r['SQ_INSTS_MFMA'],1), r['SQ_INSTS_VALU']/max(r['SQ_INSTS_MFMA'],1)-1.0, r['SQ_INSTS_LDS']/max(r['SQ_INSTS_MFMA'],1),
            100*r['SQ_LDS_BANK_CONFLICT']/max(r['SQ_LDS_IDX_ACTIVE'],1))
    print(line)
P
