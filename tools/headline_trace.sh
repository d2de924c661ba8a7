cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MMF_NOP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hlt -o p -- python3 $R/tools/profile_frames.py 200 640x480 1 1 headline > $R/gpurun_out/hlt.log 2>&1
csv=$(find $R/gpurun_out/hlt -name '*kernel_trace.csv' | head -1)
python3 $R/tools/timeline.py $csv 150 > $R/gpurun_out/hlt_timeline.txt
python3 $R/tools/timeline.py $csv 151 > $R/gpurun_out/hlt_timeline2.txt
rm -rf $R/gpurun_out/hlt
tail -3 $R/gpurun_out/hlt.log
