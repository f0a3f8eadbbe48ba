tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1; rc=$?; tail -40 gpurun_out/r04_collect.log; exit $rc
