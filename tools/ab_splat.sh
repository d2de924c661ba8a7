set -e
R=$GRAFT_REPO_ROOT
cd $R
for v in new old new old; do
  if [ $v = old ]; then export MMF_HIP_LIB=$R/ab/libmmf_old.so; else unset MMF_HIP_LIB; fi
  python bench.py --no-cpu-baseline --steps 200 > gpurun_out/ab_$v.json 2>> gpurun_out/ab.err
  python - <<PY
import json
r=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
print("$v", round(r["value"]), r["ms_per_step"], {k:round(x["us"],1) for k,x in r["surfel_passes"].items()})
PY
done
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export MMF_HIP_LIB=$R/ab/libmmf_old.so; else unset MMF_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab_$v -o p -- python3 $R/tools/profile_frames.py 200 640x480 1 1 headline > $R/gpurun_out/prof_ab_$v.log 2>&1
done
