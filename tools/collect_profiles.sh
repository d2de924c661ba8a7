#!/bin/bash
# GPU box: the artefacts profiles/README.md lists for round 2, into gpurun_out/art2/ (copy the ones to keep into profiles/)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/art2
mkdir -p $O
cd $R
python bench.py > $O/r02_bench_line.json 2> $O/bench.err || exit 1
MMF_BENCH_WORKLOAD=config5 python bench.py --no-cpu-baseline > $O/r02_bench_config5_n1.json 2>> $O/bench.err || exit 1
hipcc --offload-arch=gfx950 -O3 tools/graph_host_probe.hip -o /tmp/ghp 2>/dev/null && timeout -k 10 120 /tmp/ghp > $O/r02_graph_host_probe.txt
hipcc --offload-arch=gfx950 -O3 tools/stream_stall_probe.hip -o /tmp/ssp 2>/dev/null && timeout -k 10 120 /tmp/ssp > $O/r02_stream_stall_probe.txt
bash tools/ab_envs.sh 3 300 - MMF_GRAPHS=1 MMF_PREFETCH_EARLY=1 > $O/r02_ab_graphs_prefetch.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_headline -o p -- python3 $R/tools/profile_frames.py 120 640x480 1 1 headline > $O/prof_headline.log 2>&1
python3 $R/tools/kstats.py $(ls $O/prof_headline/*results.db | head -1) 120 track_producer rgb_step > $O/r02_kernel_stats_640x480.txt
rocprofv3 --kernel-trace --stats -d $O/prof_8m -o p -- python3 $R/tools/profile_frames.py 60 640x480 8 0 > $O/prof_8m.log 2>&1
python3 $R/tools/kstats.py $(ls $O/prof_8m/*results.db | head -1) 60 > $O/r02_kernel_stats_8models.txt
SIZE=640x480 bash $R/tools/probe_variants.sh "clean_|fuse_|index_|splat_" shipped > $O/r02_surfel_probe_fixed_map.txt 2>&1
rm -rf $O/prof_headline $O/prof_8m
ls -la $O
