# Round-4 A/B of the tree kernel on one box: in-tree library (entry-parallel mass / Jacobian assembly) vs abtest/lib_walk.so (the same
# sources with -DSP_ENTRY_PARALLEL=0: one lane per row walking its ancestors), alternating, HIP-event kernel times of bench.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; else unset DART_STEPPER_LIB; fi
  python $R/bench.py --no-extras --env-id $2 --precision $3 --envs 16384 --steps $4 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 f$3 %.3f ms (kernel %.3f)' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for rep in 1 2; do for v in ${VARIANTS:-base walk}; do run $v DartHumanWalker-v1 64 30; run $v DartHumanWalker-v1 32 40; run $v DartWalker3d-v1 64 30; run $v DartDog-v1 64 40; done; done
