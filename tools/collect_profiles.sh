#!/bin/bash
# GPU box: the artefacts profiles/README.md lists for round 4, into gpurun_out/art4/ (copy the ones to keep into profiles/)
#   /usr/local/graft/bin/gpurun --timeout 1150 -- 'bash tools/collect_profiles.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/art4
mkdir -p $O
cd $R
STAMP=$(python3 -c "import bench; print(bench.source_stamp())")
python bench.py > $O/r04_bench_line.json 2> $O/bench.err || exit 1
echo "bench line done"
cd /tmp && export TMPDIR=/tmp
MMF_BENCH_HEADLINE_ONLY=1 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1
{ echo "source_stamp $STAMP"; python3 $R/tools/kstats.py $(ls $O/prof_bench/*results.db | head -1) 1 gn_iter; } > $O/r04_bench_under_rocprofv3.txt
grep -h '"metric"' $O/prof_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('the same run, from its own JSON line: value %.0f frames/s, roofline.us_per_launch %.2f (min %.2f), gn_chain.us %.1f' % (d['value'], d['roofline']['us_per_launch'], d['roofline']['us_per_launch_min'], d['gn_chain']['us']))" >> $O/r04_bench_under_rocprofv3.txt
echo "bench under rocprofv3 done"
rocprofv3 --kernel-trace --stats -d $O/prof_headline -o p -- python3 $R/tools/profile_frames.py 400 640x480 1 1 headline > $O/prof_headline.log 2>&1
python3 $R/tools/kstats.py $(ls $O/prof_headline/*results.db | head -1) 400 gn_iter > $O/r04_kernel_stats_640x480.txt
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o p --output-format csv -- python3 $R/tools/profile_frames.py 60 640x480 1 0 headline > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o p --output-format csv -- python3 $R/tools/profile_frames.py 60 640x480 1 0 headline > $O/pmc_w.log 2>&1
python3 $R/tools/pmc_to_json.py $O/pmc_f $O/pmc_w 640 480 $O/r04_pmc_summary.json $STAMP > $O/r04_pmc_summary.txt 2>&1
echo "pmc done"
rocprofv3 --kernel-trace --stats -d $O/prof_8m -o p -- python3 $R/tools/profile_frames.py 60 640x480 8 0 > $O/prof_8m.log 2>&1
python3 $R/tools/kstats.py $(ls $O/prof_8m/*results.db | head -1) 60 > $O/r04_kernel_stats_8models.txt
python3 $R/tools/gn_iter_probe.py > $O/r04_gn_iter_probe.txt 2>&1
cd $R
bash tools/gn_floor_probe.sh > $O/r04_gn_floor_probe.txt 2>&1
python3 tools/mature_splat_probe.py > $O/r04_mature_splat_probe.txt 2>&1
python3 tools/host_frames.py > $O/r04_host_frames.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/pipeline_probe tools/pipeline_probe.hip && timeout -k 5 120 /tmp/pipeline_probe > $O/r04_pipeline_probe.txt 2>&1
bash tools/pmc_kernel.sh splat_kernel "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" > $O/r04_splat_counters.txt 2>&1
rm -rf $R/gpurun_out/pmck*
echo "probes done"
MMF_BENCH_WORKLOAD=config5 python bench.py --no-cpu-baseline --no-extras > $O/r04_bench_config5_n1.json 2>> $O/bench.err
MMF_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/r04_bench_gloo_n2.json 2>> $O/bench.err
rm -rf $O/prof_bench $O/prof_headline $O/prof_8m $O/pmc_f $O/pmc_w
ls -la $O
