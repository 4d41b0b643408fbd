#!/bin/bash
# round-5 GPU session 1: correctness of the tree, then the A/B measurements that decide which variants become the default
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s1; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu" ; (time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
run() {  # variant env precision envs steps warmup
  if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  timeout 300 python $R/bench.py --no-extras --env-id $2 --precision $3 --envs $4 --steps $5 --warmup $6 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-14s %-22s f%s  %.4f ms  %.3e' % ('$1', '$2', '$3', d['roofline']['kernel_ms'], d['value']))"
}
echo "== tree A/B"
for rep in 1 2; do for v in base r4 oplane oplane_nolicm; do
  run $v DartHumanWalker-v1 64 16384 40 5; run $v DartHumanWalker-v1 32 16384 40 5; run $v DartWalker3d-v1 64 16384 40 5; run $v DartDog-v1 64 16384 60 5
done; done 2>&1 | tee $O/ab_tree.txt
echo "== planar A/B"
for rep in 1 2; do for v in base r4 nolicm_p; do
  run $v DartHopper-v1 64 65536 2000 200; run $v DartWalker2d-v1 64 65536 500 50; run $v DartWalker2d-v1 32 65536 500 50; run $v DartHalfCheetah-v1 64 65536 100 20; run $v DartHopper-v1 32 65536 2000 200
done; done 2>&1 | tee $O/ab_planar.txt
unset DART_STEPPER_LIB
echo "== host latency"; bash $R/tools/gpu/host_latency.sh > $O/host_latency.log 2>&1; cat $R/gpurun_out/host_latency/probe.txt $R/gpurun_out/host_latency/trace_gaps.txt 2>/dev/null | tail -30
echo "== first launch"; timeout 900 bash $R/tools/gpu/first_launch.sh > $O/first_launch.log 2>&1; tail -40 $R/gpurun_out/first_launch.txt
echo "== WRITE_SIZE of the HumanWalker fp64 kernel per variant"
cd /tmp; export TMPDIR=/tmp
for v in base r4 oplane oplane_nolicm; do
  if [ "$v" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$v.so; else unset DART_STEPPER_LIB; fi
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${v}_$c -- python $R/bench.py --no-extras --precision 64 --env-id DartHumanWalker-v1 --steps 10 --warmup 2 > $O/pmc_${v}_$c.log 2>&1
  done
done
unset DART_STEPPER_LIB; cd $R
python tools/summarize_rocprof.py $O $O/pmc_summary.txt > /dev/null 2>&1; grep -i "sp_step_kernel\|^## " $O/pmc_summary.txt | head -40
find $O -name '*.db' -delete
echo "== done"
