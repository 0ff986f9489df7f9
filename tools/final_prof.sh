cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; rm -rf $O/p4; mkdir -p $O/p4;
# 1. dominant kernel: three counter passes (separate runs, kernel-trace only)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p4/f -- python3 $R/tools/roofline_kernel.py 20 > $O/p4/f.log 2>&1; echo pass1;
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p4/w -- python3 $R/tools/roofline_kernel.py 20 > $O/p4/w.log 2>&1; echo pass2;
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/p4/h -- python3 $R/tools/roofline_kernel.py 20 > $O/p4/h.log 2>&1; echo pass3;
cd $R; python tools/pmc_dominant.py $O/p4/f $O/p4/w $O/p4/h $O/pmc_dominant_kernel.json a1aabfd > $O/p4/pmc_dom.log 2>&1; tail -3 $O/p4/pmc_dom.log;
# 2. whole step: two counter passes over an eager bench run
cd /tmp; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p4/sf -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $O/p4/sf.log 2>&1; echo pass4;
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p4/sw -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $O/p4/sw.log 2>&1; echo pass5;
cd $R; python tools/hbm_by_kernel.py $O/p4/sf $O/p4/sw 90 > $O/r03_hbm_by_kernel_pmc.txt 2>&1; head -4 $O/r03_hbm_by_kernel_pmc.txt;
# 3. traces: graph mode (stats + one step) and eager with the conv trace
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/p4/graph -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/p4/graph.log 2>&1; echo graph;
GWD_TRACE_CONV=1 rocprofv3 --kernel-trace --output-format csv -d $O/p4/eager -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline > $O/p4/eager.out 2> $O/p4/eager.err; echo eager;
cd $R; f=$(ls $O/p4/graph/*/*kernel_trace.csv | head -1); python tools/trace_step.py $f 400 > $O/r03_one_step_by_category.txt; cp $(ls $O/p4/graph/*/*kernel_stats.csv | head -1) $O/r03_graph_bench_steps10_kernel_stats.csv;
python tools/conv_instep.py $O/p4/eager $O/p4/eager.err > $O/r03_conv_in_step_by_shape.txt 2>&1; python tools/conv_instep.py $O/p4/eager $O/p4/eager.err backbone > $O/r03_backbone_in_step.txt 2>&1; head -5 $O/r03_backbone_in_step.txt; head -6 $O/r03_one_step_by_category.txt;
cp $(ls $O/p4/eager/*/*kernel_trace.csv | head -1) $O/p4/eager_kt.csv; python $R/tools/lowocc.py $O/p4/eager_kt.csv > $O/r03_low_occupancy_launches.txt 2>&1; rm -rf $O/p4/f $O/p4/w $O/p4/h $O/p4/sf $O/p4/sw $O/p4/graph $O/p4/eager
# 4. SQ counters of the halo-patch kernels (forward launches of tools/convbench.py)
bash $R/tools/pmc_conv.sh "8,120,160,160,160,3;8,120,160,800,320,3" pmc_halo > /dev/null 2>&1; cat $O/pmc_halo/pmc1.txt $O/pmc_halo/pmc2.txt > $O/r03_pmc_halo_kernel.txt; python $R/tools/lowocc.py $(ls $O/p4/eager_kt.csv 2>/dev/null) > /dev/null 2>&1; echo done
