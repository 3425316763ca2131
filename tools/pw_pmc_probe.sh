#!/bin/bash
# PMC picture of the HBM-bound pool / LRN kernels in isolation at the benchmark's sizes (1024 frames): where wave time goes (issue,
# waits), VALU share, and HBM bytes (FETCH_SIZE x 2 per the guide's gfx950 correction, WRITE_SIZE) per launch.
#   bash tools/pw_pmc_probe.sh [out-tag]      (GPU box, from the repo root)  -> gpurun_out/<tag>/summary.txt
set -o pipefail
out=gpurun_out/${1:-pw_pmc}
mkdir -p $out
export TMPDIR=/tmp
for k in "pool_lrn_bwd 1" "pool_lrn_bwd 2" "lrn_pool_fwd 1" "lrn_pool_fwd 2"; do
  set -- $k
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES \
      -d $out/sq_$1_$2 -o r -- python3 $GRAFT_REPO_ROOT/tools/pw_probe.py $1 $2 1024 3 > $out/sq_$1_$2.log 2>&1 || echo "sq pass $k failed"
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS \
      -d $out/sq2_$1_$2 -o r -- python3 $GRAFT_REPO_ROOT/tools/pw_probe.py $1 $2 1024 3 > $out/sq2_$1_$2.log 2>&1 || echo "sq2 pass $k failed"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch_$1_$2 -o r -- python3 $GRAFT_REPO_ROOT/tools/pw_probe.py $1 $2 1024 3 > $out/fetch_$1_$2.log 2>&1 || echo "fetch pass $k failed"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $out/write_$1_$2 -o r -- python3 $GRAFT_REPO_ROOT/tools/pw_probe.py $1 $2 1024 3 > $out/write_$1_$2.log 2>&1 || echo "write pass $k failed"
done
python3 - "$out" <<'P' | tee $out/summary.txt
import glob, os, sqlite3, sys
out = sys.argv[1]
for k in ("pool_lrn_bwd_1", "pool_lrn_bwd_2", "lrn_pool_fwd_1", "lrn_pool_fwd_2"):
    res, dur = {}, None
    for part in ("sq", "sq2", "fetch", "write"):
        f = glob.glob("%s/%s_%s/**/*_results.db" % (out, part, k), recursive=True)
        if not f:
            continue
        c = sqlite3.connect(f[0])
        try:
            for n, cn, v, d in c.execute("select kernel_name, counter_name, avg(value), avg(duration) from counters_collection where kernel_name like '%lrn%' group by kernel_name, counter_name"):
                res[cn] = v
                if part == "sq":
                    dur = d
        except Exception as ex:
            print(k, part, "no counters:", ex)
    if not res:
        continue
    g = lambda n: res.get(n, float("nan"))
    wc = g("SQ_WAVE_CYCLES")
    print("%-16s dur %.3f ms | clock %.2f GHz | wave cycles: issuing %.0f %%, waiting to issue %.0f %%, waitcnt/barrier %.0f %% | VALU busy %.0f %% of wave cycles, %.1f VALU insts per wave-quad-cycle x1e3"
          % (k, (dur or 0) / 1e6, g("GRBM_GUI_ACTIVE") / 8 / (dur or 1), 100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_WAIT_ANY") / wc,
             100 * g("SQ_ACTIVE_INST_VALU") / wc, 1e3 * g("SQ_INSTS_VALU") / wc))
    print("                 insts per launch: VALU %.3e  LDS %.3e  VMEM rd %.3e  wr %.3e  SALU %.3e | waves %.0f | LDS-active %.0f %%  VMEM-active %.0f %%  LDS-issue-stall %.0f %% of wave cycles"
          % (g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"), g("SQ_INSTS_SALU"), g("SQ_WAVES"),
             100 * g("SQ_ACTIVE_INST_LDS") / wc, 100 * g("SQ_ACTIVE_INST_VMEM") / wc, 100 * g("SQ_WAIT_INST_LDS") / wc))
    print("                 HBM: fetched %.3f GB (FETCH_SIZE x 2: %s KB units), written %.3f GB | L2 hit %.1f %%"
          % (g("FETCH_SIZE") * 2 * 1024 / 1e9, "%.0f" % g("FETCH_SIZE"), g("WRITE_SIZE") * 1024 / 1e9, 100 * g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
P
find $out -name "*_results.db" -delete      # the summary is what is kept (gpurun copies at most 64 MiB back)
