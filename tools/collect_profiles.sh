set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --no-dp-rehearsal > $O/bench_under_rocprof.json 2>$O/trace.err &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-ml --no-dp-rehearsal > /dev/null 2>$O/f.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-ml --no-dp-rehearsal > /dev/null 2>$O/w.err &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/sq1 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-ml --no-dp-rehearsal > /dev/null 2>$O/s1.err &&
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq2 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-ml --no-dp-rehearsal > /dev/null 2>$O/s2.err
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $O/pmc_hbm_traffic.csv > $O/pmc_traffic.txt 2>&1
python tools/pmc_sq.py $O/sq1 $O/sq2 > $O/sq_counters.txt 2>&1
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python tools/stamps.py > $O/phase_stamps.txt 2>&1
python tools/dwp_phases.py >> $O/phase_stamps.txt 2>&1
python tools/dp_sim.py > $O/dp_sim.txt 2>&1
python bench.py --loss ml --no-cpu-baseline > $O/bench_ml.json 2>/dev/null
python bench.py --hidden 4096 --nhid 6 --bunch 512 --loss ml --no-cpu-baseline --steps 100 --warmup 10 > $O/bench_cfg5.json 2>/dev/null
python bench.py --bunch 256 --no-cpu-baseline --no-ml > $O/bench_b256.json 2>/dev/null
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/sq1 $O/sq2
ls -la $O
