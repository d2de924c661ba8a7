#!/bin/bash
# GPU box: the artefacts profiles/README.md lists for round 5, into gpurun_out/art5/ (copy the ones to keep into profiles/)
#   /usr/local/graft/bin/gpurun --timeout 1150 -- 'bash tools/collect_profiles.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/art5
mkdir -p $O
cd $R
STAMP=$(python3 -c "import bench; print(bench.source_stamp())")
# (the stamped summaries first, into profiles/ of this copy as well: the bench line below quotes them -- roofline.frac IS the
# rocprofv3 figure of this very command when the source stamp matches)
cd /tmp && export TMPDIR=/tmp
MMF_BENCH_HEADLINE_ONLY=1 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1
{ echo "source_stamp $STAMP"; python3 $R/tools/kstats.py $(ls $O/prof_bench/*results.db | head -1) 710 gn_iter; } > $O/r05_bench_under_rocprofv3.txt
grep -h '"metric"' $O/prof_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('the same run, from its own JSON line: value %.0f frames/s, roofline.us_per_launch %.2f (min %.2f), gn_chain.us %.1f' % (d['value'], d['roofline']['us_per_launch'], d['roofline']['us_per_launch_min'], d['gn_chain']['us']))" >> $O/r05_bench_under_rocprofv3.txt
cp $O/r05_bench_under_rocprofv3.txt $R/profiles/
echo "bench under rocprofv3 done"
rocprofv3 --kernel-trace --stats -d $O/prof_headline -o p -- python3 $R/tools/profile_frames.py 400 640x480 1 1 headline > $O/prof_headline.log 2>&1
python3 $R/tools/kstats.py $(ls $O/prof_headline/*results.db | head -1) 400 gn_iter > $O/r05_kernel_stats_640x480.txt
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o p --output-format csv -- python3 $R/tools/profile_frames.py 60 640x480 1 0 headline > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o p --output-format csv -- python3 $R/tools/profile_frames.py 60 640x480 1 0 headline > $O/pmc_w.log 2>&1
python3 $R/tools/pmc_to_json.py $O/pmc_f $O/pmc_w 640 480 $O/r05_pmc_summary.json $STAMP > $O/r05_pmc_summary.txt 2>&1
cp $O/r05_pmc_summary.json $O/r05_pmc_summary.txt $R/profiles/
echo "pmc done"
(cd $R && python bench.py 2> $O/bench.err | grep '^{"metric"' > $O/r05_bench_line.json) || exit 1
echo "bench line done"
rocprofv3 --kernel-trace --stats -d $O/prof_8m -o p -- python3 $R/tools/profile_frames.py 100 640x480 8 1 > $O/prof_8m.log 2>&1
python3 $R/tools/kstats.py $(ls $O/prof_8m/*results.db | head -1) 100 gn_iter > $O/r05_kernel_stats_8models.txt
rocprofv3 --kernel-trace --stats -d $O/prof_4m -o p -- python3 $R/tools/profile_frames.py 100 640x480 4 1 > $O/prof_4m.log 2>&1
python3 $R/tools/kstats.py $(ls $O/prof_4m/*results.db | head -1) 100 gn_iter > $O/r05_kernel_stats_4models.txt
python3 $R/tools/gn_mixed_probe.py 8 > $O/r05_gn_mixed_probe.txt 2>&1
python3 $R/tools/gn_mixed_probe.py 2 >> $O/r05_gn_mixed_probe.txt 2>&1
python3 $R/tools/mm_sparse_probe.py 8 24 > $O/r05_sparse_walk_probe.txt 2>&1
{ for m in 2 4 8; do for c in MMF_NOP=1 MMF_PASS_BATCH=0 MMF_PASS_BATCH=2 MMF_EARLY_IMAGE=start MMF_SPEC_PREP_ALL=4 MMF_GN_OBJ_FIRST=0 MMF_PREP_RECT=0 MMF_GN_MIXED_LANES=256 MMF_GN_FUSED=0; do echo "$m models $c: $(env $c timeout -k 10 200 python3 $R/tools/profile_frames.py 200 640x480 $m 1 2>&1 | grep 'ms per frame')"; done; done; } > $O/r05_multi_model_ab.txt 2>&1
(cd $R && python3 -c "import __graft_entry__ as g; g.smoke()") > $O/r05_smoke.txt 2>&1
cd $R
python3 tools/host_frames.py > $O/r05_host_frames.txt 2>&1
echo "probes done"
MMF_BENCH_WORKLOAD=config5 python bench.py --no-cpu-baseline --no-extras 2>> $O/bench.err | grep '^{"metric"' > $O/r05_bench_config5_n1.json
MMF_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>> $O/bench.err | grep '^{"metric"' > $O/r05_bench_gloo_n2.json
bash $R/tools/mm_trace.sh 8 r05tl > /dev/null 2>&1; cp $R/gpurun_out/mmt_r05tl_timeline.txt $O/r05_timeline_8models.txt; cd $R
rm -rf $O/prof_bench $O/prof_headline $O/prof_8m $O/prof_4m $O/pmc_f $O/pmc_w
ls -la $O
