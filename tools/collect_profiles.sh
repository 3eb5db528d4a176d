# Regenerates the per-round evidence under gpurun_out/r04prof on the GPU box (copy what is to be judged into profiles/).
#   bash tools/collect_profiles.sh a   -- rocprofv3 passes over the default workload (kernel trace + stats, PMC)
#   bash tools/collect_profiles.sh b   -- un-profiled bench lines, in-kernel stamps, dp_sim, the other configs
#   bash tools/collect_profiles.sh c   -- rocprofv3 passes over the config-5 shape (k_fwd64 / k_dx64 / k_dwp<8>)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-kernel-timing --no-ml --no-dp-rehearsal --budget-s 2000"
C5="--hidden 4096 --nhid 6 --bunch 512 --loss ml"
case "$1" in
a)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --no-dp-rehearsal --budget-s 2000 > $O/bench_under_rocprof.json 2>$O/trace.err &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 40 --warmup 10 $Q > /dev/null 2>$O/f.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 40 --warmup 10 $Q > /dev/null 2>$O/w.err &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/sq1 -- python3 $R/bench.py --steps 20 --warmup 5 $Q > /dev/null 2>$O/s1.err &&
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq2 -- python3 $R/bench.py --steps 20 --warmup 5 $Q > /dev/null 2>$O/s2.err
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $O/pmc_hbm_traffic.csv > $O/pmc_traffic.txt 2>&1
python tools/pmc_sq.py $O/sq1 $O/sq2 > $O/sq_counters.txt 2>&1
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/sq1 $O/sq2
;;
b)
cd $R
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python tools/stamps.py > $O/phase_stamps.txt 2>&1
python tools/dwp_phases.py >> $O/phase_stamps.txt 2>&1
python tools/dp_sim.py > $O/dp_sim.txt 2>&1
python bench.py --loss ml --no-cpu-baseline --no-dp-rehearsal > $O/bench_ml.json 2>/dev/null
python bench.py $C5 --no-cpu-baseline --no-dp-rehearsal --steps 100 --warmup 10 > $O/bench_cfg5.json 2>/dev/null
python bench.py --bunch 256 --no-cpu-baseline --no-ml --no-dp-rehearsal > $O/bench_b256.json 2>/dev/null
python bench.py --bunch 512 --no-cpu-baseline --no-ml --no-dp-rehearsal > $O/bench_b512.json 2>/dev/null
;;
c)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace5 -- python3 $R/bench.py $C5 --steps 40 --warmup 10 --windows 3 $Q > /dev/null 2>$O/t5.err &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/sq1_5 -- python3 $R/bench.py $C5 --steps 8 --warmup 2 --windows 2 $Q > /dev/null 2>$O/s15.err &&
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq2_5 -- python3 $R/bench.py $C5 --steps 8 --warmup 2 --windows 2 $Q > /dev/null 2>$O/s25.err &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch5 -- python3 $R/bench.py $C5 --steps 8 --warmup 2 --windows 2 $Q > /dev/null 2>$O/f5.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write5 -- python3 $R/bench.py $C5 --steps 8 --warmup 2 --windows 2 $Q > /dev/null 2>$O/w5.err
cd $R
python tools/pmc_sq.py $O/sq1_5 $O/sq2_5 > $O/sq_counters_cfg5.txt 2>&1
python tools/pmc_traffic.py $O/pmc_fetch5 $O/pmc_write5 $O/pmc_traffic_cfg5.json $O/pmc_hbm_traffic_cfg5.csv > $O/pmc_traffic_cfg5.txt 2>&1
find $O/trace5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_cfg5.csv
rm -rf $O/trace5 $O/sq1_5 $O/sq2_5 $O/pmc_fetch5 $O/pmc_write5
;;
esac
ls -la $O
